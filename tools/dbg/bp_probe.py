"""Development (GPU box, library built with -DFMD_X_PROBE: tools/build_variant.sh xprobe "-DFMD_X_PROBE"): cycles k_extract_bp's wavefronts 0 and 3
spend in the phases of a tile (a wavefront owns a tile: no barriers between them), averaged over every 61st workgroup of the bench's 4096-station blocks."""
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
names = ["window split into bf16 halves, span cubics asked for", "the workgroup's prologue (tables, estimates, S_old, barriers), per tile", "FIRs (88 ds_read_b128, 104 v_mfma)", "outputs from the accumulators, stores", "-"]
for w, off in (("wavefront 0", 0), ("wavefront 3", 8)):
    print(w, "per tile", round(sum(v[off:off + 5]) / cnt))
    for i, nme in enumerate(names[:4]):
        print(f"   {nme:72s} {v[off + i] / cnt:8.0f} cycles")
print("sampled tiles:", cnt)
out2 = (C.c_ulonglong * 16)()
if dm.L.fmd_debug_read_x_probe2(out2) == 0:
    w = list(out2)
    print("workgroup lifetime per tile %.0f cycles = %.2f us: shader clock %.0f MHz under this load" % (w[7] / cnt, w[15] / cnt / 100.0, 100.0 * w[7] / max(w[15], 1)))
