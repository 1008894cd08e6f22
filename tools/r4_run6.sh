#!/bin/bash
O=gpurun_out/r4_7; mkdir -p $O
./tools/permlane_probe > $O/permlane.txt 2>&1
python -m pytest tests/test_gpu_fast.py tests/test_gpu_scale.py tests/test_gpu_long.py -m gpu -q 2>&1 | tail -40 > $O/tests_fast.log
bash tools/r4_ab_pv.sh "pv4 pv3" > $O/ab.log 2>&1
