#!/bin/bash
# Development tool (GPU box): bench.py value / ms per step of library variants tools/ab/*.so on the same box, interleaved.
# usage: tools/ab_bench.sh "variant names" [rounds] [bench args]
L=fm-radio_amd/csrc/libfmdemod.so
cp $L /tmp/orig.so
export GPU_MAX_HW_QUEUES=8
for r in $(seq 1 ${2:-2}); do for v in $1; do
  cp tools/ab/$v.so $L
  python bench.py ${3:---fast-math} --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})" $v
done; done
cp /tmp/orig.so $L
