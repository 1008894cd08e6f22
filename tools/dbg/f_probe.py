"""Development (GPU box, library built with -DFMD_F_PROBE: tools/build_variant.sh xprobe "-DFMD_F_PROBE"): cycles the front end (k_front_mfma) spends
in each barrier-separated phase and waiting at each barrier, averaged over every 61st workgroup of the bench's 4096-station blocks (five workgroups per CU)."""
import sys, ctypes as C, numpy as np, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import torch, fmradio_loader, synth
pkg = fmradio_loader.load()
n_ch, bs, nb = 4096, 16384, 12
dm = pkg.BatchDemod(n_ch, bs, 256_000, fast_math=True)
base = np.stack([synth.to_cf32(synth.fm_capture(nb * bs, fs=256000.0, seed=6300, channel=c)["iq"]) for c in range(2)])
dbase = torch.from_numpy(base).cuda()
idx = torch.from_numpy(np.arange(n_ch) % 2).cuda()
blocks = [dbase[:, b * bs:(b + 1) * bs][idx].contiguous() for b in range(nb)]
for b in range(nb):
    dm.submit(blocks[b])
dm.synchronize()
out = (C.c_ulonglong * 16)()
assert dm.L.fmd_debug_read_f_probe(out) == 0
v = list(out); cnt = max(v[7], 1)
names = ["loads + arctangent", "wait barrier 1", "phase differences, split", "wait barrier 2 + stores of the halves", "wait barrier 3", "decimating FIR (MFMA), column sums, stores"]
print("wavefront 0 of the front end's workgroups, total", round(sum(v[0:6]) / cnt))
for i, nme in enumerate(names):
    print(f"   {nme:44s} {v[i] / cnt:8.0f} cycles")
print("sampled workgroups:", cnt)
