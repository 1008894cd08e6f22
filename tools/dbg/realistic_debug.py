"""Development (GPU box): where the tolerance mode leaves the oracle on a realistic capture — per station the L+R error, the number of
direction-antipodal consecutive u8 samples (ties of the discriminator's wrap), and the raw samples around the largest fm_out differences.
    python tools/dbg/realistic_debug.py [condition] [stations] [blocks]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, R + "/tests", R + "/oracle"): sys.path.insert(0, p)
import numpy as np
import fmradio_loader, oraclelib as O, synth
from gpu_parity import run_gpu, lib_coeffs_to_oracle
pkg = fmradio_loader.load()
cond = sys.argv[1] if len(sys.argv) > 1 else "fading_5hz"
NCH = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 156
fs, bs = 256000, 16384
kw = synth.REALISTIC_CONDITIONS[cond]
seed = 7000 + 37 * sorted(synth.REALISTIC_CONDITIONS).index(cond)
caps = np.stack([synth.to_u8(synth.fm_capture_realistic(nb * bs, fs=float(fs), seed=seed, channel=c, **kw)["iq"]) for c in range(NCH)])
g = run_gpu(pkg, caps, bs, fs, fast_math=True)
for c in range(NCH):
    o = O.run_chain(caps[c], bs, fs, u8=True, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["fm_out_iq", "lpr", "lmr"])
    xy = caps[c].astype(np.int64) - 127
    x, y = xy[:, 0], xy[:, 1]
    cross = x[1:] * y[:-1] - y[1:] * x[:-1]; dot = x[1:] * x[:-1] + y[1:] * y[:-1]
    tie = np.nonzero((cross == 0) & (dot < 0))[0] + 1
    fo_g = np.asarray(g["fm_out_iq"][c], np.float64).reshape(-1, 2)[:, 0]; fo_o = o["fm_out_iq"].reshape(-1, 2)[:, 0].astype(np.float64)
    d = fo_g - fo_o
    e_lpr = np.sqrt(np.mean((np.asarray(g["lpr"][c], np.float64) - o["lpr"]) ** 2))
    big = np.nonzero(np.abs(d) > 1e-3)[0]
    print(f"station {c}: lpr rms err {e_lpr:.2e}; ties {tie.size}; zero samples {int(((x == 0) & (y == 0)).sum())}; fm_out samples off by > 1e-3: {big.size}; fm_out rms err {np.sqrt(np.mean(d ** 2)):.2e}")
    if big.size:
        # clusters of differing fm_out samples -> the input sample they come from (fm_out[k] ~ input 2 k - 32 - 63 .. 2 k - 32: delayed by the analytic signal's 32)
        starts = big[np.concatenate([[True], np.diff(big) > 64])]
        for s in starts[:6]:
            i0 = 2 * (s - 32) - 70
            near = tie[(tie >= i0 - 8) & (tie <= i0 + 160)]
            seg = xy[max(i0, 0) + 60: max(i0, 0) + 76]
            print(f"   fm_out[{s}] diff {d[s]:+.3f} ... ties nearby (input index): {near[:5]}; raw (x, y) around: {[tuple(int(v) for v in r) for r in seg]}")
