#!/bin/bash
O=gpurun_out/r4_5; mkdir -p $O
Q="--no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
for rep in 1 2; do
python bench.py $Q > $O/bench_fused_$rep.json 2> $O/bench.err
FMD_NO_FUSED_PLL=1 python bench.py $Q > $O/bench_unfused_$rep.json 2>> $O/bench.err
FMD_PLL_EAGER=1 python bench.py $Q > $O/bench_eager_$rep.json 2>> $O/bench.err
done
python bench.py $Q --steps 20 --warmup 5 > $O/bench_driver.json 2>> $O/bench.err
python bench.py $Q --no-kernel-times > $O/bench_nokt.json 2>> $O/bench.err
python bench.py $Q --fs 1024000 > $O/bench_1024k.json 2>> $O/bench.err
FMD_PLL_EAGER=1 python bench.py $Q --fs 1024000 > $O/bench_1024k_eager.json 2>> $O/bench.err
python bench.py $Q --u8 > $O/bench_u8.json 2>> $O/bench.err
python bench.py $Q --channels 8192 > $O/bench_8192ch.json 2>> $O/bench.err
python -m pytest tests/test_gpu_fast.py tests/test_gpu_long.py tests/test_gpu_scale.py -m gpu -q 2>&1 | tail -8 > $O/tests_fast.log
bash tools/r3_alone_trace.sh > $O/alone.log 2>&1
