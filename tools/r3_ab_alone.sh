#!/bin/bash
# Development tool (GPU box): two library variants (tools/ab/*.so) on the same box — the pipelined step and every throughput stage
# on its own (FMD_DEBUG_SKIP_STAGES; stage bits 1 front, 8 PLL, 16 extract, 32 RDS).
L=fm-radio_amd/csrc/libfmdemod.so
cp $L /tmp/orig.so
export GPU_MAX_HW_QUEUES=8
run() { FMD_DEBUG_SKIP_STAGES=$2 python bench.py $3 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1"; }
for r in 1 2; do for v in ${1:-old new}; do
  cp tools/ab/$v.so $L
  run "$v all" 0 "$2"; run "$v front" 56 "$2"; run "$v extract" 41 "$2"; run "$v pll" 49 "$2"; run "$v no-rds" 32 "$2"
done; done
cp /tmp/orig.so $L
