#!/bin/bash
O=gpurun_out/r4_11; mkdir -p $O
Q="--no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
python bench.py $Q > $O/bench_fused.json 2> $O/bench.err
FMD_NO_FUSED_PLL=1 python bench.py $Q > $O/bench_unfused.json 2>> $O/bench.err
FMD_PLL_EAGER=1 python bench.py $Q > $O/bench_eager.json 2>> $O/bench.err
bash tools/r3_alone_trace.sh > $O/alone.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fusedtrace -- python3 bench.py --no-kernel-times $Q > /dev/null 2>&1
f=$(find /tmp/fusedtrace -name "*kernel_stats.csv" | head -1); cp $f $O/fused_kernel_stats.csv
FMD_PLL_EAGER=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eagertrace -- python3 bench.py --no-kernel-times $Q > /dev/null 2>&1
f=$(find /tmp/eagertrace -name "*kernel_stats.csv" | head -1); cp $f $O/eager_kernel_stats.csv
