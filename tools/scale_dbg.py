import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import torch, fmradio_loader, synth
from gpu_parity import run_gpu
pkg = fmradio_loader.load()
n_ch, bs = 4096, 16384
base = np.stack([synth.to_cf32(synth.fm_capture(2 * bs, fs=256000.0, seed=900, channel=c)["iq"]) for c in range(8)])
small = run_gpu(pkg, base, bs, 256_000)
idx = np.arange(n_ch) % 8
for trial in range(3):
    dm = pkg.BatchDemod(n_ch, bs, 256_000, keep_taps=True, pipelined=(len(sys.argv) < 2))
    for b in range(2):
        blk = torch.from_numpy(np.ascontiguousarray(base[:, b * bs:(b + 1) * bs])).cuda()[torch.from_numpy(idx).cuda()].contiguous()
        dm.process(blk)
    dt = dm.stream("pll_dt")
    ref = small["pll_dt"][:, -dt.shape[1]:]
    bad = np.nonzero((dt.view(np.uint32) != ref[idx].view(np.uint32)).any(axis=1))[0]
    print("trial", trial, "bad channels", len(bad), bad[:20], "first bad sample", [int(np.nonzero(dt[c].view(np.uint32) != ref[idx[c]].view(np.uint32))[0][0]) for c in bad[:8]])
    if len(bad):
        c = bad[0]
        w = np.nonzero(dt[c].view(np.uint32) != ref[idx[c]].view(np.uint32))[0]
        print("  channel", c, "bad samples", len(w), w[:40], "values", dt[c][w[:6]], "expected", ref[idx[c]][w[:6]])
        c = bad[-1]
        w = np.nonzero(dt[c].view(np.uint32) != ref[idx[c]].view(np.uint32))[0]
        print("  channel", c, "bad samples", len(w), w[:40])
    dm.close()
