#!/bin/bash
# Development A/B (GPU box) through the dev build's environment switches (make -C fm-radio_amd/csrc dev -> tools/ab/dev.so).
# usage: tools/ab_env.sh "NAME1=VAR=1 NAME2=" rounds "bench args"     (an empty VAR list = the default schedule)
L=fm-radio_amd/csrc/libfmdemod.so; cp $L /tmp/orig.so; cp tools/ab/dev.so $L
O=gpurun_out/ab_env; mkdir -p $O
for r in $(seq 1 ${2:-2}); do for spec in $1; do
  name=${spec%%=*}; var=${spec#*=}
  ( [ -n "$var" ] && export $var; python bench.py --no-cpu-baseline --no-other-mode --no-configs --no-host-fed $3 2>/dev/null ) | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})" $name "$3" | tee -a $O/table.txt
done; done
cp /tmp/orig.so $L
