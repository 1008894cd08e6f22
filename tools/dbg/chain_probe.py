"""Development (GPU box, library built with -DFMD_C_PROBE: tools/build_variant.sh cprobe "-DFMD_C_PROBE"): where k_chain's wavefronts spend their
cycles — wavefront 0 of the team and the pilot wavefront of every 61st workgroup, per phase, averaged per half-step.  usage: chain_probe.py [stations]"""
import sys, ctypes as C, numpy as np, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import torch, fmradio_loader, synth
pkg = fmradio_loader.load()
n_ch, bs, nb = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 16384, 12
dm = pkg.BatchDemod(n_ch, bs, 256_000, fast_math=True)
dm.set_chain(True)
base = np.stack([synth.to_cf32(synth.fm_capture(nb * bs, fs=256000.0, seed=6300, channel=c)["iq"]) for c in range(2)])
dbase = torch.from_numpy(base).cuda()
idx = torch.from_numpy(np.arange(n_ch) % 2).cuda()
blocks = [dbase[:, b * bs:(b + 1) * bs][idx].contiguous() for b in range(nb)]
for b in range(nb):
    dm.submit(blocks[b])
dm.synchronize()
out = (C.c_ulonglong * 32)()
assert dm.L.fmd_debug_read_c_probe(out) == 0
v = list(out); wgs = max(v[13], 1); hs = wgs * 17.0
team = ["(loop overhead / X prologue)", "D of the half-step's first station (8 arctangents a thread) + loads issued", "D halves written", "wait at barrier A", "M: matrix products, window + column sums written",
        "D of the next station beside it + loads issued", "wait at barrier B", "X: FIRs + outputs of a station-tile"]
print("team wavefront 0, cycles per half-step (17 per block):")
for i, nme in enumerate(team):
    print(f"   {nme:84s} {v[i] / hs:8.0f}")
print(f"   {'sum':84s} {sum(v[0:8]) / hs:8.0f}")
pil = ["(loop overhead)", "tile's inputs from LDS, first span's phase", "its share of D", "spans (8 a half-step)", "wait at barrier A", "wait at barrier B"]
print("pilot wavefront, cycles per half-step:")
for i, nme in enumerate(pil):
    print(f"   {nme:84s} {v[16 + i] / hs:8.0f}")
print(f"   {'sum':84s} {sum(v[16:22]) / hs:8.0f}")
print("workgroup lifetime %.0f cycles = %.1f us per block: shader clock %.0f MHz; %d workgroup-blocks sampled" % (v[14] / wgs, v[15] / wgs / 100.0, 100.0 * v[14] / max(v[15], 1), wgs))

st = (C.c_ulonglong * 2048)()
if dm.L.fmd_debug_read_c_start(st) == 0:
    a = np.array(list(st), np.float64).reshape(-1, 2)[: max(1, n_ch // 8)]
    t0 = a[:, 0].min()
    s_us, e_us = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    print("newest launch: %d workgroups; started within 5 us of the first: %d; start times (us) percentiles 50/90/100: %.0f %.0f %.0f; launch length %.0f us; lifetimes 50/100: %.0f %.0f us"
          % (a.shape[0], int((s_us < 5).sum()), np.percentile(s_us, 50), np.percentile(s_us, 90), s_us.max(), e_us.max(), np.percentile(e_us - s_us, 50), (e_us - s_us).max()))
