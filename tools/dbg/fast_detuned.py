import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "tests", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import fmradio_loader, oraclelib as O, synth
from gpu_parity import run_gpu, lib_coeffs_to_oracle
pkg = fmradio_loader.load(); pkg.load_library()
nb = 10; n = nb * 16384
kinds = [dict(seed=501, channel=0), dict(seed=502, channel=1, pilot_hz=19130.0), dict(seed=504, channel=3, pilot_level=0.02, noise_sigma=0.3), dict(seed=505, channel=4, pilot_hz=18870.0), dict(seed=503, channel=2, pilot_level=0.0)]
caps = np.stack([synth.to_cf32(synth.fm_capture(n, fs=256_000.0, **k)["iq"]) for k in kinds])
order = [int(a) for a in sys.argv[1:]] or list(range(5))
caps = caps[order]
g = run_gpu(pkg, caps, 16384, 256_000, fast_math=True)
for c in range(len(order)):
    o = O.run_chain(caps[c], 16384, 256_000, u8=False, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["pll_dt", "audio", "lmr"])
    d = np.asarray(g["pll_dt"][c], np.float64).reshape(nb, -1) - o["pll_dt"].reshape(nb, -1); d -= np.round(d)
    a = np.asarray(g["audio"][c], np.float64).reshape(nb, -1) - o["audio"].reshape(nb, -1)
    print("capture", order[c], "pll_dt rms", " ".join(f"{np.sqrt(np.mean(d[i]**2)):.1e}" for i in range(nb)))
    print("          audio rms ", " ".join(f"{np.sqrt(np.mean(a[i]**2)):.1e}" for i in range(nb)))
    big = np.argwhere(np.abs(d) > 1e-4)
    if big.size:
        b, i = big[0]
        print("          first |d|>1e-4 at block", b, "sample", i, "d around:", d[b, max(0, i - 3):i + 4])
if len(order) == 1:
    d = np.asarray(g["pll_dt"][0], np.float64).reshape(nb, -1) - o["pll_dt"].reshape(nb, -1); d -= np.round(d)
    np.set_printoptions(linewidth=200, precision=2)
    print("d[0, :130] * 1e8:\n", (d[0, :130] * 1e8).round(1))
    print("gpu dt[0,:6]", np.asarray(g["pll_dt"][0])[:6], "oracle", o["pll_dt"][:6])
