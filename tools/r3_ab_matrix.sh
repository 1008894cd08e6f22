#!/bin/bash
# Development tool (GPU box): two library variants (tools/ab/*.so) over the bench workloads, and under a per-block consumer.
L=fm-radio_amd/csrc/libfmdemod.so
cp $L /tmp/orig.so
export GPU_MAX_HW_QUEUES=8
run() { python bench.py $2 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" "$1 $2"; }
for a in "" "--u8" "--unlocked-frac 0.25 --unlocked-kind mix" "--channels 1024" "--channels 2048" "--channels 8192" "--fs 1024000" "--fs 1024000 --u8" "--fs 2048000 --channels 2048"; do
  for v in $1; do cp tools/ab/$v.so $L; run "$v" "$a"; done
done
for v in $1; do cp tools/ab/$v.so $L; echo "$v per-block consumer:"; WAIT=1 python3 tools/host_submit_times.py 2>&1 | grep total; echo "$v free-running (noise input):"; python3 tools/host_submit_times.py 2>&1 | grep total; done
cp /tmp/orig.so $L
