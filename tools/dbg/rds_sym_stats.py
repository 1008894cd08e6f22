"""Development tool (GPU box): per-station statistics of the tolerance mode's soft RDS symbols against the oracle (the setup of
tests/test_gpu_long.py::test_tolerance_mode_rds_stage_on_pipelined_wavefronts)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import fmradio_loader
import test_gpu_long as TL
from concurrent.futures import ProcessPoolExecutor
pkg = fmradio_loader.load()
n_st, n_blocks, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 32, 6600
dm = pkg.BatchDemod(n_st, TL.BS, TL.FS, keep_taps=False, fast_math=True)
coeff = bytes(dm.get_coeffs(0))
with ProcessPoolExecutor(min(n_st, 32)) as ex:
    res = sorted(ex.map(TL._oracle_rds, [(n_blocks, seed, c, coeff) for c in range(n_st)]), key=lambda r: r[0])
caps = np.stack([r[1] for r in res])
syms = [[] for _ in range(n_st)]; counts = np.zeros((n_st, n_blocks), np.int64)
for b in range(n_blocks):
    assert dm.process(np.ascontiguousarray(caps[:, b * TL.BS:(b + 1) * TL.BS])) == 0
    sy, sc = dm.rds_symbols()
    for c in range(n_st):
        syms[c].append(sy[c, :sc[c]].copy()); counts[c, b] = sc[c]
dm.close()
rows = []
for c in range(n_st):
    o_cnt = np.asarray(res[c][4]).reshape(-1)[:n_blocks]
    g = np.concatenate(syms[c]).astype(np.float64); o = np.asarray(res[c][3], np.float64).reshape(-1)
    same = np.array_equal(counts[c], o_cnt)
    first_diff = int(np.argmax(counts[c] != o_cnt)) if not same else -1
    # align from the first block after which the counts agree for good
    ok_from = 0
    for b in range(n_blocks - 1, -1, -1):
        if counts[c, b] != o_cnt[b]:
            ok_from = b + 1; break
    lo_g, lo_o = int(counts[c, :max(ok_from, 8)].sum()), int(o_cnt[:max(ok_from, 8)].sum())
    m = min(g.size - lo_g, o.size - lo_o)
    d = np.abs(g[lo_g:lo_g + m] - o[lo_o:lo_o + m])
    rows.append((c, same, first_diff, ok_from, m, float(np.median(d)), float(np.percentile(d, 99)), float(np.sqrt(np.mean(d ** 2))), float(d.max())))
    print(f"station {c:3d} counts_equal={same} first_diff_block={first_diff:3d} agree_from={ok_from:3d} n={m:5d} median {rows[-1][5]:.2e} p99 {rows[-1][6]:.2e} rms {rows[-1][7]:.2e} max {rows[-1][8]:.2e}")
