#!/bin/bash
# Development tool (GPU box): the tolerance mode's first decimator on the matrix cores (k_predecim_mfma) against the VALU form
# (FMD_PREDECIM_VALU=1) at 1.024 and 2.048 MSa/s: the pipelined step and the decimator alone (FMD_DEBUG_SKIP_STAGES 63).
export GPU_MAX_HW_QUEUES=16
run() { python bench.py $2 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1"; }
for a in "--fs 1024000" "--fs 1024000 --u8" "--fs 2048000 --channels 2048" "--fs 2048000 --channels 2048 --u8"; do
  run "mfma all $a" "$a"; FMD_DEBUG_SKIP_STAGES=63 run "mfma alone $a" "$a"
  FMD_PREDECIM_VALU=1 run "valu all $a" "$a"; FMD_PREDECIM_VALU=1 FMD_DEBUG_SKIP_STAGES=63 run "valu alone $a" "$a"
done
