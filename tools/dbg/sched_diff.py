"""Development tool (GPU box): per-block difference of the tolerance mode's audio between schedules (unpipelined reference, pipelined
process, deferred submit)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import fmradio_loader, synth, torch
pkg = fmradio_loader.load()
n_ch, bs, nb = 3072, 16384, 7
base = np.stack([synth.to_cf32(synth.fm_capture(nb * bs, fs=256000.0, seed=6300, channel=c)["iq"]) for c in range(4)])
idx = torch.from_numpy(np.arange(n_ch) % 4).cuda()
dbase = torch.from_numpy(base).cuda()
blocks = [dbase[:, b * bs:(b + 1) * bs][idx].contiguous() for b in range(nb)]
def run(mode):
    kw = dict(fast_math=True)
    if mode == "unpipelined": kw["pipelined"] = False
    dm = pkg.BatchDemod(n_ch, bs, 256000, **kw)
    out = []
    if mode in ("unpipelined", "process"):
        for b in range(nb):
            dm.process(blocks[b]); out.append(dm.audio().copy())
    else:
        dm.set_output_lag(True)
        side = torch.cuda.Stream()
        kept = {}
        for b in range(nb):
            dm.submit(blocks[b])
            with torch.cuda.stream(side):
                dm.wait_outputs(side)
                if b >= 1:
                    kept[b - 1] = dm.audio_tensor().clone(); dm.release_outputs(side)
        dm.synchronize(); side.synchronize()
        out = [kept[b].cpu().numpy() for b in range(nb - 1)] + [dm.audio().copy()]
    dm.close()
    return out
ref = run("unpipelined")
for mode in sys.argv[1:]:
    got = run(mode)
    print(mode, os.environ.get("FMD_NO_FUSED_PLL"), os.environ.get("FMD_PLL_EAGER"), " ".join(f"{np.abs(g.astype(np.float64) - r).max():.2e}" for g, r in zip(got, ref)))
