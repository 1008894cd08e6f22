"""Development tool: one 4096-station handle vs two 2048-station handles fed alternately (independent pipelines on one GPU)."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch, fmradio_loader
pkg = fmradio_loader.load()
import bench
C, block, fs = 4096, 16384, 256000
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = 100
dev = torch.device('cuda', 0)
x = bench.synth_block_device(torch, C, 8 * block, float(fs), 1234, dev, False).view(C, 8, block, 2).permute(1, 0, 2, 3).contiguous()
cp = C // parts
xs = [[x[b, i * cp:(i + 1) * cp].contiguous() for i in range(parts)] for b in range(8)]
dms = [pkg.BatchDemod(cp, block, fs, device=0) for _ in range(parts)]
def run(n):
    for k in range(n):
        for i, dm in enumerate(dms):
            dm.process(xs[k % 8][i])
    for dm in dms: dm.synchronize()
    torch.cuda.synchronize()
run(24)
res = []
for rep in range(4):
    t0 = time.perf_counter(); run(steps); res.append(1e3 * (time.perf_counter() - t0) / steps)
print("%d handle(s) x %d stations: ms/step %s (min %.3f) -> %.0f MSa/s" % (parts, cp, " ".join("%.3f" % r for r in res), min(res), C * block / min(res) / 1e3))
