"""Development tool (GPU box): tolerance mode vs the oracle, per stream and block, with where the largest difference sits."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, R + "/tests", R + "/oracle"): sys.path.insert(0, p)
import numpy as np
import fmradio_loader, oraclelib as O, synth
from gpu_parity import run_gpu, lib_coeffs_to_oracle
pkg = fmradio_loader.load()
fs, bs, nb = 256000, 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 6
NCH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
caps = np.stack([synth.to_cf32(synth.fm_capture(nb * bs, fs=float(fs), seed=9100, channel=c)["iq"]) for c in range(NCH)])
g = run_gpu(pkg, caps, bs, fs, fast_math=True)
for c in range(NCH):
    o = O.run_chain(caps[c], bs, fs, u8=False, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["fm_out_iq", "pll_dt", "lpr", "lmr", "rds", "audio", "rds_sym", "lmr_phase"])
    for k in ("pll_dt", "lmr", "rds", "audio", "lmr_phase"):
        a = np.asarray(g[k][c], np.float64).reshape(nb, -1); b = o[k].reshape(nb, -1).astype(np.float64)
        d = a - b
        if k == "pll_dt": d -= np.round(d)
        per = np.sqrt((d ** 2).mean(axis=1))
        i = np.unravel_index(np.argmax(np.abs(d)), d.shape)
        print(c, k, "rms per block:", " ".join("%.1e" % x for x in per), "| max %.2e at block %d idx %d of %d; signal rms %.2e" % (abs(d[i]), i[0], i[1], d.shape[1], np.sqrt((b ** 2).mean())))
    print(c, "counts", g["rds_count"][c], o["rds_count"])
