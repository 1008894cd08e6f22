"""Per-block, per-stream error of the fast mode against the oracle (debugging aid)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "tests", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import fmradio_loader, oraclelib as O, synth
from gpu_parity import run_gpu, lib_coeffs_to_oracle
pkg = fmradio_loader.load(); pkg.load_library()
fs = int(sys.argv[1]) if len(sys.argv) > 1 else 256000
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bs = fs * 64 // 1000; nb = 12
caps = np.stack([synth.to_cf32(synth.fm_capture(nb * bs, fs=float(fs), seed=9100, channel=c)["iq"]) for c in range(nch)])
g = run_gpu(pkg, caps, bs, fs, fast_math=True)
m = fs // 256000; nfo = bs // m // 2
for c in range(nch):
    o = O.run_chain(caps[c], bs, fs, u8=False, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["fm_out_iq", "pll_dt", "lpr", "lmr", "audio", "rds_sym", "rds", "lmr_phase", "rds_raw_sym"])
    print("channel", c, "counts equal", np.array_equal(g["rds_count"][c], o["rds_count"]), "bytes equal", np.array_equal(g["rds_bytes"][c], o["rds_bytes"]), g["rds_count"][c], o["rds_count"])
    per = {"fm_out_iq": 2 * nfo, "pll_dt": nfo, "lpr": nfo // 4, "lmr": nfo // 4, "audio": nfo // 2, "rds": nfo // 4}
    for k, w in per.items():
        a = np.asarray(g[k][c], np.float64).reshape(nb, w); b = o[k].reshape(nb, w).astype(np.float64)
        d = a - b
        if k == "pll_dt":
            d -= np.round(d)
        print(f"  {k:10s}", " ".join(f"{np.sqrt(np.mean(d[i]**2)):.1e}" for i in range(nb)), " sig rms", f"{np.sqrt(np.mean(b[-1]**2)):.2e}")
    if np.array_equal(g["rds_count"][c], o["rds_count"]):
        dd = g["rds_sym"][c].astype(np.float64) - o["rds_sym"]
        cs = np.concatenate([[0], np.cumsum(o["rds_count"])])
        print("  rds_sym   ", " ".join(f"{np.sqrt(np.mean(dd[cs[i]:cs[i+1]]**2)):.1e}" for i in range(nb)), " sig rms", f"{np.sqrt(np.mean(o['rds_sym']**2)):.2e}")
    d = np.asarray(g['pll_dt'][c], np.float64).reshape(nb, nfo) - o['pll_dt'].reshape(nb, nfo); d -= np.round(d)
    print("  pll_dt mean diff", " ".join(f"{np.mean(d[i]):+.1e}" for i in range(nb)))
    print("  lmr_phase gpu", g["lmr_phase"][c][-2:], "oracle", o["lmr_phase"][-2:])
    if c == 0:
        blk = 8
        dd = d[blk]
        print("  pll_dt diff, block 8, first 192 samples (x1e6):", np.array2string(dd[:192] * 1e6, precision=0, max_line_width=220))
