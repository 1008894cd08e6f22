#!/bin/bash
# Development tool (GPU box): two library variants (tools/ab/*.so) on the same box — the pipelined step at 4096 / 2048 / 1024 stations
# and the RDS stage on its own (FMD_DEBUG_SKIP_STAGES 25: front, PLL and extract skipped).
L=fm-radio_amd/csrc/libfmdemod.so
cp $L /tmp/orig.so
export GPU_MAX_HW_QUEUES=16
run() { FMD_DEBUG_SKIP_STAGES=$2 python bench.py $3 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1"; }
for r in 1 2; do for v in ${1:-rds3x rds4}; do
  cp tools/ab/$v.so $L
  run "$v 4096" 0 ""; run "$v 2048" 0 "--channels 2048"; run "$v 1024" 0 "--channels 1024"; run "$v rds-alone" 25 ""
done; done
cp /tmp/orig.so $L
