"""Development tool: host time of every submit call and (under rocprofv3) the kernel sequence of a stage-skipping run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, fmradio_loader, bench
pkg = fmradio_loader.load()
C, block, fs = 4096, 16384, 256000
dev = torch.device('cuda', 0)
x = bench.synth_block_device(torch, C, 8 * block, float(fs), 1234, dev, False).view(C, 8, block, 2).permute(1, 0, 2, 3).contiguous()
dm = pkg.BatchDemod(C, block, fs, device=0, fast_math=True)
for k in range(24): dm.submit(x[k % 8])
dm.synchronize(); torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for k in range(60):
    a = time.perf_counter(); dm.submit(x[k % 8]); ts.append((a - t0, time.perf_counter() - a))
dm.synchronize(); torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("ms/step %.4f" % (tot / 60 * 1e3))
print("host: call start (us) / call duration (us):", " ".join("%d/%d" % (a * 1e6, d * 1e6) for a, d in ts))
