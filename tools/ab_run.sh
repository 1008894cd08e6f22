#!/bin/bash
# Development tool (GPU box): steady-state ms/step of library variants tools/ab/*.so on the same box, interleaved.
# usage: tools/ab_run.sh "A B" "4096 1024" [rounds] [fs]
L=fm-radio_amd/csrc/libfmdemod.so
cp $L /tmp/orig.so
for r in $(seq 1 ${3:-2}); do for C in $2; do for v in $1; do
  cp tools/ab/$v.so $L; echo -n "$v C=$C: "; python tools/host_submit_probe.py $C $((400000 / C > 200 ? 200 : 400000 / C)) ${4:-256000} | sed 's/.*repeats: //'
done; done; done
cp /tmp/orig.so $L
