#!/bin/bash
export GPU_MAX_HW_QUEUES=16
run() { python bench.py $2 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1"; }
for r in 1 2; do
run "t2048 all" "--u8"; FMD_DEBUG_SKIP_STAGES=126 run "t2048 front alone" "--u8"
FMD_FRONT_U8_T1024=1 run "t1024 all" "--u8"; FMD_FRONT_U8_T1024=1 FMD_DEBUG_SKIP_STAGES=126 run "t1024 front alone" "--u8"
done
