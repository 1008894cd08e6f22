#!/bin/bash
# round 2, GPU call 1: full GPU test-suite, baseline bench (driver-style), unlocked-channel sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2_1
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2_1/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_1/bench_default.json 2> gpurun_out/r2_1/bench_default.err
for p in 0.01 0.1 1.0; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --unlocked-frac $p > gpurun_out/r2_1/bench_unlocked_$p.json 2> gpurun_out/r2_1/bench_unlocked_$p.err
done
for k in nopilot noise zero; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --unlocked-frac 0.1 --unlocked-kind $k > gpurun_out/r2_1/bench_unlocked_0.1_$k.json 2> gpurun_out/r2_1/bench_unlocked_0.1_$k.err
done
tail -5 gpurun_out/r2_1/pytest.log
for f in gpurun_out/r2_1/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(round(d['value']), d['ms_per_step'], d['roofline']['kernels_ms_per_step'] if d.get('roofline') else None, d['speculation'])
except Exception as e: print('ERR', e)
PY
done
