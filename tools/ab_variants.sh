#!/bin/bash
# Development A/B (GPU box): library variants tools/ab/<name>.so (tools/build_variant.sh), interleaved: the step, and the front end's kernel on its own
# (FMD_DEBUG_SKIP_STAGES=56).  usage: tools/ab_variants.sh "v1 v2" rounds "bench args"
L=fm-radio_amd/csrc/libfmdemod.so; cp $L /tmp/orig.so
O=gpurun_out/ab_variants; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-mode --no-configs --no-host-fed $3"
for r in $(seq 1 ${2:-2}); do for v in $1; do
  cp tools/ab/$v.so $L
  a=$($B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})")
  b=$(FMD_DEBUG_SKIP_STAGES=56 $B --no-kernel-times 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('front end alone', round(d['ms_per_step'],4))")
  echo "$v [$3] $a $b" | tee -a $O/table.txt
done; done
cp /tmp/orig.so $L
