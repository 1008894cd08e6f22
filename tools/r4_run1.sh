#!/bin/bash
# Round-4 GPU run: the tolerance-mode tests, then the default bench line and the stages alone (k_pll_sparse in place of k_pll_span)
O=gpurun_out/r4_1; mkdir -p $O
python -m pytest tests/test_gpu_fast.py tests/test_gpu_long.py -m gpu -q -x -s 2>&1 | tail -60 > $O/tests_fast.log
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fast.py --deselect tests/test_gpu_long.py 2>&1 | tail -15 > $O/tests_rest.log
Q="--no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
python bench.py $Q > $O/bench.json 2> $O/bench.err
python bench.py $Q --steps 20 --warmup 5 > $O/bench_driver.json 2>> $O/bench.err
bash tools/r3_alone_trace.sh > $O/alone.log 2>&1
