#!/bin/bash
# Development A/B (GPU box): bench.py --wideband with library variants tools/ab/<name>.so, interleaved.  usage: tools/ab_wideband.sh "v1 v2" rounds
L=fm-radio_amd/csrc/libfmdemod.so; cp $L /tmp/orig.so
O=gpurun_out/ab_wideband; mkdir -p $O
for r in $(seq 1 ${2:-2}); do for v in $1; do
  cp tools/ab/$v.so $L
  python bench.py --wideband --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'step ms', round(d['ms_per_step'],4), 'x real time', round(d['realtime_factor'],1), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})" $v | tee -a $O/table.txt
done; done
cp /tmp/orig.so $L
