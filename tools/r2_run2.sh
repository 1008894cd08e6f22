#!/bin/bash
# round 2, GPU call 2: fast-math mode correctness + first timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2_2
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_fast.py -m gpu -q -s 2>&1 | tail -60 > gpurun_out/r2_2/pytest_fast.log
for extra in "" "--unlocked-frac 0.1" "--unlocked-frac 1.0" "--no-pipeline" "--channels 8192" "--channels 16384" "--fs 1024000" "--u8"; do
  tag=$(echo "$extra" | tr -d ' -.')
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $extra > gpurun_out/r2_2/bench_fast_$tag.json 2> gpurun_out/r2_2/bench_fast_$tag.err
done
cat gpurun_out/r2_2/pytest_fast.log
for f in gpurun_out/r2_2/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(round(d['value']), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()} if d.get('roofline') else None, d['speculation'])
except Exception as e: print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-2000:])
PY
done
