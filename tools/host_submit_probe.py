"""Development tool: steady-state ms/step of the pipelined library (no event bracketing), several repeats."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch, fmradio_loader
pkg = fmradio_loader.load()
import bench
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fs = int(sys.argv[3]) if len(sys.argv) > 3 else 256000
block = 16384 * (fs // 256000)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda', 0)
NB = 8 if fs == 256000 else 2
x = bench.synth_block_device(torch, C, NB * block, float(fs), 1234, dev, False).view(C, NB, block, 2).permute(1, 0, 2, 3).contiguous()
dm = pkg.BatchDemod(C, block, fs, device=0)
for k in range(24): dm.process(x[k % NB])
dm.synchronize(); torch.cuda.synchronize()
res = []
for rep in range(4):
    t0 = time.perf_counter()
    for k in range(steps): dm.process(x[k % NB])
    dm.synchronize(); torch.cuda.synchronize()
    res.append(1e3 * (time.perf_counter() - t0) / steps)
print("ms/step over %d steps, 4 repeats: %s  (min %.3f)" % (steps, " ".join("%.3f" % r for r in res), min(res)))
