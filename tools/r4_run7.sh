#!/bin/bash
O=gpurun_out/r4_8; mkdir -p $O
python tools/dbg/sched_diff.py process submit > $O/diff.log 2>&1
FMD_NO_FUSED_PLL=1 python tools/dbg/sched_diff.py submit >> $O/diff.log 2>&1
FMD_PLL_EAGER=1 python tools/dbg/sched_diff.py submit >> $O/diff.log 2>&1
bash tools/r4_ab_pv.sh "pv5 pv3" > $O/ab.log 2>&1
