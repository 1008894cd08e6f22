"""Development (GPU box): the tolerance mode's outputs of a short run, to compare two builds of the library bit for bit.
usage: python tools/dbg/lib_dump.py out.npz [fs]   (run once per build, then: python tools/dbg/lib_dump.py --cmp a.npz b.npz)"""
import sys, os
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))]
    print("identical" if not bad else "DIFFERENT: %s" % bad)
    for k in bad:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)); print(" ", k, "max abs diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape))
    sys.exit(1 if bad else 0)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import fmradio_loader, synth
from gpu_parity import run_gpu
pkg = fmradio_loader.load()
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 256000
bs = fs * 64 // 1000
caps = np.stack([synth.to_cf32(synth.fm_capture(8 * bs, fs=float(fs), seed=7700, channel=c)["iq"]) for c in range(5)])
g = run_gpu(pkg, caps, bs, fs, fast_math=True)
out = {k: v for k, v in g.items() if isinstance(v, np.ndarray)}
out["rds_bytes"] = np.concatenate([np.asarray(x) for x in g["rds_bytes"]])
out["rds_sym"] = np.concatenate([np.asarray(x) for x in g["rds_sym"]])
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], sorted(out))
