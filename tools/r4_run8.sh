#!/bin/bash
O=gpurun_out/r4_9; mkdir -p $O
python tools/dbg/sched_diff2.py > $O/diff.log 2>&1
