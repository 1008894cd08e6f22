#!/bin/bash
# Development A/B (GPU box): bench line of library variants tools/ab/*.so, interleaved.  usage: tools/r4_ab_bench.sh "v1 v2" [rounds] [bench args]
L=fm-radio_amd/csrc/libfmdemod.so; cp $L /tmp/orig.so
O=gpurun_out/r4_ab; mkdir -p $O
for r in $(seq 1 ${2:-2}); do for v in $1; do
  cp tools/ab/$v.so $L
  python bench.py --no-cpu-baseline --no-other-mode --no-configs --no-host-fed $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})" $v | tee -a $O/table.txt
done; done
cp /tmp/orig.so $L
