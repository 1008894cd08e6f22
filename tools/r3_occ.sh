#!/bin/bash
# Development tool (GPU box): k_front_mfma's occupancy limited through padded dynamic LDS (FMD_FRONT_LDS_PAD): alone and in the pipeline.
export GPU_MAX_HW_QUEUES=8
run() { FMD_FRONT_LDS_PAD=$1 FMD_DEBUG_SKIP_STAGES=$2 python bench.py --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "pad=$1 skip=$2"; }
for p in 0 13000 24000 32000 45000 72000; do run $p 56; run $p 0; run $p 40; done
