#!/bin/bash
# quick look on the GPU box: the default (tolerance) and exact bench lines, twice
export GPU_MAX_HW_QUEUES=8
run() { python bench.py $2 --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})" "$1"; }
for r in 1 2; do
  run fast ""
  run exact "--exact"
done
for a in "$@"; do run "fast $a" "$a"; done
