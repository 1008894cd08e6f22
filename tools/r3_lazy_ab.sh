#!/bin/bash
# Development tool (GPU box): tolerance-mode step with the deferred extract stage (default) and without (FMD_NO_LAZY_EXTRACT=1: the extract
# stage on its own stream beside the next front end), over a few workloads.
export GPU_MAX_HW_QUEUES=8
run() { env $1 python bench.py $2 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1 $2"; }
for a in "" "--u8" "--unlocked-frac 0.25 --unlocked-kind mix" "--unlocked-frac 1.0 --unlocked-kind noise" "--channels 1024" "--channels 2048" "--channels 8192" "--fs 1024000" "--fs 1024000 --u8"; do
  run "X=1" "$a"; run "FMD_NO_LAZY_EXTRACT=1" "$a"
done
