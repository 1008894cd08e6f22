#!/bin/bash
O=gpurun_out/r4_10; mkdir -p $O
python tools/dbg/sched_diff2.py > $O/diff.log 2>&1
python tools/dbg/sched_diff.py process submit >> $O/diff.log 2>&1
FMD_NO_FUSED_PLL=1 python tools/dbg/sched_diff.py submit >> $O/diff.log 2>&1
python -m pytest tests/test_gpu_fast.py tests/test_gpu_scale.py tests/test_gpu_long.py -m gpu -q 2>&1 | tail -12 > $O/tests_fast.log
bash tools/r4_ab_pv.sh "pv5" > $O/ab.log 2>&1
