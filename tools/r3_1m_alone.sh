#!/bin/bash
# Development tool (GPU box): the 1.024 MSa/s configuration — the pipelined step and every stage on its own
# (FMD_DEBUG_SKIP_STAGES; stage bits 1 front, 8 PLL, 16 extract, 32 RDS, 64 first decimator).
export GPU_MAX_HW_QUEUES=16
run() { FMD_DEBUG_SKIP_STAGES=$2 python bench.py --fs 1024000 $3 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1"; }
for a in "" "--u8"; do
  run "all $a" 0 "$a"; run "predecim $a" 63 "$a"; run "front $a" 126 "$a"; run "pll $a" 119 "$a"; run "extract $a" 111 "$a"; run "rds $a" 95 "$a"; run "no-predecim $a" 64 "$a"; run "predecim+front $a" 62 "$a"
done
