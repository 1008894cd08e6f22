#!/bin/bash
# Everything a round's profiles/ directory is made from, in one gpurun call (GPU box, repo root): counters and traces of the default line
# (tools/collect_round.sh), the driver's command, the stages alone (development build of the library: tools/ab/dev.so, built by
# `make -C fm-radio_amd/csrc dev`) and the measurement table.  ROUND=N (default 5).
export ROUND=${ROUND:-5}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r${ROUND}_evidence; mkdir -p $O
bash tools/collect_round.sh > $O/collect.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
if [ -f tools/ab/dev.so ]; then
  L=fm-radio_amd/csrc/libfmdemod.so; cp $L /tmp/ship.so; cp tools/ab/dev.so $L
  bash tools/alone_trace.sh > $O/alone.log 2>&1
  bash tools/bounds.sh > $O/bounds.log 2>&1
  cp /tmp/ship.so $L
fi
bash tools/measure_table.sh > $O/table.log 2>&1
cp gpurun_out/table/table.jsonl $O/measurement_table.jsonl 2>/dev/null
