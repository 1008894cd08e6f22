#!/bin/bash
# A/B on one box: the step with and without the k_lmr_phase launch behind k_extract (timing only: results are wrong without it)
export GPU_MAX_HW_QUEUES=8
run() { python bench.py $2 --steps 60 --warmup 5 --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})" "$1"; }
for r in 1 2; do
  run base_fast ""
  FMD_DEBUG_SKIP_LMR=1 run skip_fast ""
  run base_exact "--exact"
  FMD_DEBUG_SKIP_LMR=1 run skip_exact "--exact"
done
run base_fast_nokt "--no-kernel-times"
FMD_DEBUG_SKIP_LMR=1 run skip_fast_nokt "--no-kernel-times"
