#!/usr/bin/env python3
"""Development probe (GPU box): host-side duration of every fmd_submit_cf32_dev call of a run of resident blocks (tolerance mode,
4096 stations), and when each returns relative to the first — is the host ever the one the GPU waits for?"""
import os
import sys
import time
import pathlib
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import fmradio_loader
pkg = fmradio_loader.load()
pkg.load_library()
import torch

C, N, FS = int(os.environ.get("CH", "4096")), 16384, 256000
dev = torch.device("cuda:0")
x = torch.randn(2, C, N, 2, device=dev) * 0.3
dm = pkg.BatchDemod(C, N, FS, device=0, fast_math=True)
WAIT = os.environ.get("WAIT") == "1"      # a consumer that orders a stream behind every block's outputs (the multi-GPU gather does)
LAG = os.environ.get("LAG") == "1"        # ... one submission later (fmd_set_output_lag)
if LAG:
    dm.set_output_lag(True)
gs = torch.cuda.Stream(device=dev)
stage = torch.empty(C, N // 8, 2, dtype=torch.int16, device=dev)
for k in range(24):
    dm.submit(x[k % 2])
    if WAIT and (k >= 1 or not LAG):
        dm.audio_pcm16_into(stage, gs)
dm.synchronize()
ts = []
t0 = time.perf_counter()
for k in range(40):
    a = time.perf_counter()
    dm.submit(x[k % 2])
    if WAIT:
        dm.audio_pcm16_into(stage, gs)
    ts.append((a - t0, time.perf_counter() - a))
dm.synchronize()
tot = time.perf_counter() - t0
print("total ms/step", round(tot / 40 * 1e3, 4))
print("submit durations us:", [round(d * 1e6) for _, d in ts])
print("submit start times us:", [round(a * 1e6) for a, _ in ts])
dm.close()
