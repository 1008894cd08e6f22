#!/bin/bash
O=gpurun_out/r4_12; mkdir -p $O
Q="--no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
python -m pytest tests/test_gpu_fast.py tests/test_gpu_scale.py -m gpu -q 2>&1 | tail -5 > $O/tests_fast.log
for rep in 1 2; do
python bench.py $Q > $O/bench_fused_$rep.json 2> $O/bench.err
FMD_PLL_EAGER=1 python bench.py $Q > $O/bench_eager_$rep.json 2>> $O/bench.err
done
python bench.py $Q --steps 20 --warmup 5 > $O/bench_driver.json 2>> $O/bench.err
bash tools/r3_alone_trace.sh > $O/alone.log 2>&1
