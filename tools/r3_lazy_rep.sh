#!/bin/bash
export GPU_MAX_HW_QUEUES=8
run() { env $1 python bench.py $2 --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1 $2"; }
for i in 1 2 3; do run "X=1" ""; run "X=1" "--no-kernel-times"; run "FMD_NO_LAZY_EXTRACT=1" "--no-kernel-times"; done
