#!/usr/bin/env python3
"""Development probe (GPU box): the exact mode's pipelined step as the first handle of a process and as the second (the HIP runtime
maps streams onto GPU_MAX_HW_QUEUES hardware queues in creation order: which stages share a queue depends on what was created before)."""
import os
import sys
import time
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import fmradio_loader
pkg = fmradio_loader.load()
pkg.load_library()
import torch
sys.path.insert(0, str(ROOT))
import bench

C, N, FS = 4096, 16384, 256000
dev = torch.device("cuda:0")
x = bench.synth_block_device(torch, C, 8 * N, float(FS), 1234, dev, False)
x = x.view(C, 8, N, 2).permute(1, 0, 2, 3).contiguous()


def run(dm, steps=40, pre=24):
    for k in range(pre):
        dm.submit(x[k % 8])
    dm.synchronize()
    t0 = time.perf_counter()
    for k in range(pre, pre + steps):
        dm.submit(x[k % 8])
    dm.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


n_dummy = int(os.environ.get("DUMMY_STREAMS", "0"))
dummies = [torch.cuda.Stream(device=dev) for _ in range(n_dummy)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()
a = pkg.BatchDemod(C, N, FS, device=0, fast_math=False)
print(f"{n_dummy} streams created first: exact handle {run(a):.4f} ms/step")
if os.environ.get("BISECT"):
    print(f"   again: {run(a, pre=0):.4f}")
    a.spec_stats(reset=True)
    print(f"   after spec_stats(reset=True): {run(a, pre=0):.4f}")
    a.profile(0)
    print(f"   after profile(0): {run(a, pre=0):.4f}")
    print(f"   100 steps: {run(a, steps=100, pre=0):.4f}")
    c = pkg.BatchDemod(C, N, FS, device=0, pipelined=True, pll_kernel="auto", fast_math=False)
    print(f"   another handle, 16 + 2 preroll, 100 steps: {run(c, steps=100, pre=18):.4f}")
    c.close()
b = pkg.BatchDemod(C, N, FS, device=0, fast_math=False)
print(f"   a second exact handle beside it: {run(b):.4f} ms/step")
a.close(); b.close()
