#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace statistics of the default bench command plus the two
# HBM-traffic counter passes (separate --pmc runs, no tracing domains combined with them).  Outputs under gpurun_out/prof/.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=8
X="${BENCH_ARGS:-}"      # e.g. BENCH_ARGS=--exact for the exact mode (default: the tolerance mode, bench.py's default)
python3 bench.py $X > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py $X --no-cpu-baseline --no-other-mode --no-configs > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py $X --steps 4 --warmup 1 --preroll 16 --no-cpu-baseline --no-other-mode --no-configs --no-pipeline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py $X --steps 4 --warmup 1 --preroll 16 --no-cpu-baseline --no-other-mode --no-configs --no-pipeline > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_insts -- python3 bench.py $X --steps 4 --warmup 1 --preroll 16 --no-cpu-baseline --no-other-mode --no-configs --no-pipeline > /dev/null 2> $O/pmc_insts.err
find $O -name "*.csv" | head -20
