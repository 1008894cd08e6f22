#!/bin/bash
# Development tool (GPU box): ms/step per PLL kernel choice and batch size (thresholds in fmd_api.cpp).
LIST=${1:-"2048 3072 4096 6144 8192"}
for C in $LIST; do for k in time_parallel time_parallel8 low_work; do
  echo -n "C=$C $k: "; python3 bench.py --no-cpu-baseline --no-kernel-times --channels $C --pll-kernel $k --steps $((C > 4096 ? 50 : 100)) 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.0f MSa/s' % (d['ms_per_step'], d['value']))"
done; done
