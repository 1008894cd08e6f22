#!/bin/bash
export GPU_MAX_HW_QUEUES=8
run() { python bench.py $2 --steps 60 --warmup 5 --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})" "$1"; }
for r in 1 2; do
  run base_fast ""
  FMD_DEBUG_DEFER=1 run defer_fast ""
done
run base_16k "--channels 16384"
FMD_DEBUG_DEFER=1 run defer_16k "--channels 16384"
run base_2k "--channels 2048"
FMD_DEBUG_DEFER=1 run defer_2k "--channels 2048"
