#!/bin/bash
# Development tool (GPU box): the pipelined tolerance-mode step under different stream priorities (FMD_STREAM_PRIORITIES=front,pll,extract,rds).
export GPU_MAX_HW_QUEUES=8
run() { FMD_STREAM_PRIORITIES=$1 python bench.py $2 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/tmp/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1 $2"; grep -m1 "priority range" /tmp/err.log; }
for r in 1 2; do
for p in "0,0,0,0" "0,-1,0,-1" "1,-1,0,-1" "1,-1,0,0" "0,-1,-1,-1" "1,0,0,0" "0,0,-1,0" "1,-1,-1,-1"; do run $p "$1"; done
done
