#!/usr/bin/env python3
"""Development probe (GPU box): the pipelined step with the system's HIP runtime (libfmdemod.so loaded first: /opt/rocm) and with the one
PyTorch bundles (torch imported first) — the process uses whichever libamdhip64.so.7 is loaded first.  ORDER=lib|torch, MODE=fast|exact."""
import os, sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
order, mode = os.environ.get("ORDER", "torch"), os.environ.get("MODE", "fast")
if order == "lib":
    import fmradio_loader
    pkg = fmradio_loader.load(); pkg.load_library()
    import torch
else:
    import torch
    import fmradio_loader
    pkg = fmradio_loader.load(); pkg.load_library()
import bench
dev = torch.device("cuda", 0)
C, N, FS = int(os.environ.get("CH", "4096")), 16384, 256000
x = bench.synth_block_device(torch, C, 8 * N, float(FS), 1234, dev, os.environ.get("U8") == "1")
x = x.view(C, 8, N, 2).permute(1, 0, 2, 3).contiguous()
torch.cuda.synchronize()      # (submit() does not order itself behind torch's stream: the synthesis must have finished)
dm = pkg.BatchDemod(C, N, FS, device=0, fast_math=(mode == "fast"))
P, K = int(os.environ.get("PRE", "21")), int(os.environ.get("STEPS", "100"))
for k in range(P):
    dm.submit(x[k % 8])
dm.synchronize()
t0 = time.perf_counter()
for k in range(P, P + K):
    dm.submit(x[k % 8])
dm.synchronize()
el = (time.perf_counter() - t0) / K
print(f"runtime of {order:5s} mode {mode:5s} C={C}: {el * 1e3:.4f} ms/step  {C * N / el / 1e6:.0f} MSa/s")
dm.close()
