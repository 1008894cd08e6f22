import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import fmradio_loader; pkg = fmradio_loader.load()
import test_gpu_fast as T
import oraclelib as O
from gpu_parity import lib_coeffs_to_oracle, run_gpu
for bs, nb, ch in ((65536, 4, 1), (65536, 8, 1), (16384, 16, 1)):
    caps = T._caps(2, nb * bs, 256_000.0, seed=9300)[ch:ch+1]
    g = run_gpu(pkg, caps, bs, 256_000, fast_math=True)
    o = O.run_chain(caps[0], bs, 256_000, u8=False, coeffs=lib_coeffs_to_oracle(g["coeffs"][0]), streams=["lmr", "lmr_phase", "pll_dt", "lpr"])
    print(bs, "gpu lmr_phase", np.asarray(g["lmr_phase"][0]).reshape(-1)[:nb])
    print(bs, "ora lmr_phase", o["lmr_phase"].reshape(-1)[:nb])
    n_a = bs // 8
    a = np.asarray(g["lmr"][0], np.float64).reshape(nb, n_a); b = o["lmr"].reshape(nb, n_a).astype(np.float64)
    print(bs, "lmr rms err per block", np.sqrt(np.mean((a - b) ** 2, axis=1)), "lmr rms", np.sqrt(np.mean(b ** 2)))
    d = np.asarray(g["pll_dt"][0], np.float64).reshape(-1) - o["pll_dt"].reshape(-1); d -= np.round(d)
    print(bs, "pll_dt err mean per block", d.reshape(nb, -1).mean(axis=1), "rms", np.sqrt((d**2).mean()))
