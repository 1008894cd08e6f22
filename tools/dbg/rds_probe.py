"""Development (GPU box, library built with -DFMD_RDS_PROBE: tools/build_variant.sh rdsprobe "-DFMD_RDS_PROBE"): where k_rds_sync3's wavefronts spend their cycles."""
import sys, ctypes as C, importlib, numpy as np, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import torch, fmradio_loader, synth
pkg = fmradio_loader.load()
n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bs = 16384
dm = pkg.BatchDemod(n_ch, bs, 256_000, fast_math=True)
base = np.stack([synth.to_cf32(synth.fm_capture(20 * bs, fs=256000.0, seed=6300, channel=c)["iq"]) for c in range(2)])
dbase = torch.from_numpy(base).cuda()
idx = torch.from_numpy(np.arange(n_ch) % 2).cuda()
for b in range(20):
    dm.submit(dbase[:, b * bs:(b + 1) * bs][idx].contiguous())
dm.synchronize()
out = (C.c_ulonglong * 16)()
assert dm.L.fmd_debug_read_rds_probe(out) == 0
v = list(out)
names = {0: "B2 dump", 1: "loader", 2: "A mixer", 3: "B1 clock"}
groups = bs // 16 // 4 + 2
for r in range(4):
    print(f"{names[r]:10s} work {v[2 * r] / groups:8.1f} cycles/group   barrier wait {v[2 * r + 1] / groups:8.1f}")
print("total cycles", v[8], "ticks(100MHz)", v[9], "=> core MHz", round(v[8] / max(v[9], 1) * 100), " us", v[9] / 100, " cycles/group", round(v[8] / groups))
print("prologue cycles per role (B2, loader, A, B1):", v[10:14], " B2 epilogue cycles:", v[15])
