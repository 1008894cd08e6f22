#!/bin/bash
# Bounding experiment (GPU box; needs the development build of the library: make -C fm-radio_amd/csrc dev): the tolerance-mode step with single stages not launched (FMD_DEBUG_SKIP_STAGES,
# outputs are garbage) — what each serial stage costs the pipelined step.  Stage bits: 8 = PLL, 16 = extract, 32 = RDS.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/bounds; mkdir -p $O; rm -f $O/table.jsonl
run() { FMD_DEBUG_SKIP_STAGES=$2 python bench.py $3 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'cfg': sys.argv[1], 'value': round(d['value']), 'ms': round(d['ms_per_step'],4)}))" "$1" | tee -a $O/table.jsonl; }
for rep in 1 2; do
run "all" 0 ""
run "no rds" 32 ""
run "no pll" 8 ""
run "no rds, no pll" 40 ""
run "no extract, no rds" 48 ""
run "front only" 56 ""
run "no pll no extract" 24 ""
done
run "1024 all" 0 "--channels 1024"
run "1024 no rds" 32 "--channels 1024"
run "1024 no rds no pll" 40 "--channels 1024"
run "8192 all" 0 "--channels 8192"
run "8192 no rds" 32 "--channels 8192"
run "u8 all" 0 "--u8"
run "u8 no rds" 32 "--u8"
