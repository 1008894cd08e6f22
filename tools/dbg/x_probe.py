"""Development (GPU box, library built with -DFMD_X_PROBE: tools/build_variant.sh xprobe "-DFMD_X_PROBE"): cycles k_extract_mfma's wavefronts 0 and 3 spend
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
assert dm.L.fmd_debug_read_x_probe(out) == 0
v = list(out); cnt = max(v[7], 1)
names = ["loads + split (phase 0)", "wait barrier 1", "Hilbert + NCO phases (phase 1)", "wait barrier 2", "mixers (phase 2)", "wait barrier 3", "FIRs (phase 3)"]
for w, off in (("wavefront 0", 0), ("wavefront 3", 8)):
    print(w, "total", round(sum(v[off:off + 7]) / cnt))
    for i, nme in enumerate(names):
        print(f"   {nme:34s} {v[off + i] / cnt:8.0f} cycles")
print("sampled workgroups:", cnt)
out2 = (C.c_ulonglong * 16)()
if hasattr(dm.L, "fmd_debug_read_x_probe2") and dm.L.fmd_debug_read_x_probe2(out2) == 0:
    w = list(out2)
    for nme, off in (("wavefront 0", 0), ("wavefront 3", 8)):
        print(nme, "inside phase 1: Hilbert tile(s) %.0f, NCO phases %.0f, L-R offset %.0f cycles" % (w[off] / cnt, w[off + 1] / cnt, w[off + 2] / cnt))
    print("wavefront 0 inside phase 3: the station's image slot %.0f, first product (image + LDS operands there) %.0f, the other 17 products + result store %.0f; barrier 4 + epilogue (both wavefronts) %.0f / %.0f" % (w[3] / cnt, w[4] / cnt, w[5] / cnt, w[6] / cnt, w[14] / cnt))
    print("workgroup lifetime %.0f cycles = %.2f us: shader clock %.0f MHz under this load" % (w[7] / cnt, w[15] / cnt / 100.0, 100.0 * w[7] / max(w[15], 1)))
