#!/bin/bash
# Development tool (GPU box): bench.py step of library variants tools/ab/*.so, interleaved.  usage: tools/r3_ab_simple.sh "v1 v2" [bench args]
L=fm-radio_amd/csrc/libfmdemod.so
cp $L /tmp/orig.so
export GPU_MAX_HW_QUEUES=8
for r in 1 2; do for v in $1; do
  cp tools/ab/$v.so $L
  python bench.py $2 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" $v
done; done
cp /tmp/orig.so $L
