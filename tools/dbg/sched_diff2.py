"""Development tool (GPU box): one and two blocks through fmd_submit_* against fmd_process_*: the pilot stage's cubics and the audio."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import fmradio_loader, synth, torch
pkg = fmradio_loader.load()
n_ch, bs, nb = 3072, 16384, 3
base = np.stack([synth.to_cf32(synth.fm_capture(nb * bs, fs=256000.0, seed=6300, channel=c)["iq"]) for c in range(4)])
idx = torch.from_numpy(np.arange(n_ch) % 4).cuda()
dbase = torch.from_numpy(base).cuda()
blocks = [dbase[:, b * bs:(b + 1) * bs][idx].contiguous() for b in range(nb)]
def run(mode, n):
    dm = pkg.BatchDemod(n_ch, bs, 256000, fast_math=True)
    for b in range(n):
        (dm.process if mode == "process" else dm.submit)(blocks[b])
    dm.synchronize()
    out = (dm.audio().copy(), dm.stream("pll_poly").copy(), dm.stream("rds").copy())
    dm.close()
    return out
for n in (1, 2, 3):
    a = run("process", n); b = run("submit", n)
    print("blocks", n, "audio diff", np.abs(a[0] - b[0]).max(), "poly diff", np.abs(a[1] - b[1]).max(), "rds diff", np.abs(a[2] - b[2]).max(),
          "stations with poly diff", int(np.sum(np.abs(a[1] - b[1]).reshape(n_ch, -1).max(axis=1) > 0)), "first", np.flatnonzero(np.abs(a[1] - b[1]).reshape(n_ch, -1).max(axis=1) > 0)[:8])
