#!/bin/bash
# Development tool (GPU box): the current library with and without an environment switch — pipelined step and the stages alone.
# usage: tools/r3_env_ab.sh VAR [bench args]
export GPU_MAX_HW_QUEUES=8
run() { env $4 FMD_DEBUG_SKIP_STAGES=$2 python bench.py $3 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1"; }
for r in 1 2; do for v in "X=0" "$1=1"; do
  run "$v all" 0 "$2" $v; run "$v front" 56 "$2" $v; run "$v extract" 41 "$2" $v; run "$v no-rds" 32 "$2" $v
done; done
