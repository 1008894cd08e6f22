#!/bin/bash
# Development tool (GPU box): the pipelined step against the number of hardware queues the HIP runtime may open (GPU_MAX_HW_QUEUES).
for q in 4 8 12 16 24; do for v in $1; do
  cp tools/ab/$v.so fm-radio_amd/csrc/libfmdemod.so
  GPU_MAX_HW_QUEUES=$q python bench.py $2 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$v q=$q"
done; done
