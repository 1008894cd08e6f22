"""Development: the exact mode's pilot-PLL kernel block by block — spans, samples per span, spans run in the sequence form, serial chunks — on
eight synthetic stations (one wavefront of the 8-lane kernel): all locked; seven locked and one without pilot / all-zero / detuned / noise only.
   python tools/dbg/pll_forms.py [time_parallel8|time_parallel]"""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import fmradio_loader, synth
pkg = fmradio_loader.load(); pkg.load_library()
kern = sys.argv[1] if len(sys.argv) > 1 else "time_parallel8"
nb, C = 10, 8
def cap(c, **kw):
    return synth.to_cf32(synth.fm_capture(nb * 16384, fs=256_000.0, seed=23 + c, channel=c, **kw)["iq"])
odd = {"locked": None, "no pilot": dict(pilot_level=0.0), "zero": "zero", "signed zero": "szero", "detuned": dict(pilot_hz=19130.0), "noise": "noise"}
only = sys.argv[2] if len(sys.argv) > 2 else None
for name, kw in odd.items():
    if only and name != only: continue
    rows = [cap(c) for c in range(C)]
    if kw == "zero": rows[3] = np.zeros_like(rows[3])
    elif kw == "szero": rows[3] = np.where(np.random.default_rng(6).random(rows[3].shape) < 0.5, np.float32(0.0), np.float32(-0.0)).astype(np.float32)   # (bench.py's dead channel: 0 x a signed factor)
    elif kw == "noise": rows[3] = (0.02 * np.random.default_rng(5).standard_normal(rows[3].shape)).astype(np.float32)
    elif kw is not None: rows[3] = cap(3, **kw)
    caps = np.stack(rows)
    dm = pkg.BatchDemod(n_channels=C, block_size=16384, fs_baseband=256_000, pll_kernel=kern)
    for b in range(nb):
        dm.process(caps[:, b * 16384:(b + 1) * 16384])
        p = dm.spec_stats(reset=True)["pll"]
        print(name, b, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in p.items() if k != "chunks"})
    dm.close()
