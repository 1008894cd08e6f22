"""Development (GPU box): where a station's audio depends on the batch it is in (the tap-table form of the extract stage).
    FMD_BP_TAB=1 python tools/dbg/tab_debug.py [n_ch]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, R + "/tests", R + "/oracle"): sys.path.insert(0, p)
import numpy as np, torch
import fmradio_loader
from gpu_parity import run_gpu
from test_gpu_fast import _caps
pkg = fmradio_loader.load()
bs, nb = 16384, 3
n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 1027
base = _caps(4, nb * bs, 256_000.0, seed=9300)
small = run_gpu(pkg, base, bs, 256_000, fast_math=True)
idx = np.arange(n_ch) % 4
dbase = torch.from_numpy(base).cuda(); tidx = torch.from_numpy(idx).cuda()
dm = pkg.BatchDemod(n_ch, bs, 256_000, fast_math=True)
for b in range(nb):
    dm.process(dbase[:, b * bs:(b + 1) * bs][tidx].contiguous())
    audio = dm.audio()
    want = small["audio"].reshape(4, -1, 2)[:, b * audio.shape[1]:(b + 1) * audio.shape[1]]
    diff = audio != want[idx]
    st = np.nonzero(diff.any(axis=(1, 2)))[0]
    print(f"block {b}: stations differing {st.size} of {n_ch}; first {st[:8]}")
    for c in st[:4]:
        where = np.nonzero(diff[c].any(axis=1))[0]
        print(f"   station {c}: {where.size} frames, first {where[:10]} last {where[-3:]}, max |d| {np.abs(audio[c] - want[idx][c]).max():.3e}")
dm.close()
