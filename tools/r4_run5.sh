#!/bin/bash
O=gpurun_out/r4_6; mkdir -p $O
Q="--no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
for rep in 1 2; do
python bench.py $Q > $O/bench_fused_$rep.json 2> $O/bench.err
FMD_NO_FUSED_PLL=1 python bench.py $Q > $O/bench_unfused_$rep.json 2>> $O/bench.err
FMD_PLL_EAGER=1 python bench.py $Q > $O/bench_eager_$rep.json 2>> $O/bench.err
done
bash tools/r3_alone_trace.sh > $O/alone.log 2>&1
python -m pytest tests/test_gpu_fast.py tests/test_gpu_scale.py -m gpu -q -x 2>&1 | tail -40 > $O/tests_fast.log
