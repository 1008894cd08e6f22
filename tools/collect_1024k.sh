#!/bin/bash
# GPU box: kernel-trace statistics of the 1.024 MSa/s configurations (cf32 and u8); fmd:: kernels only -> gpurun_out/prof1024/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof1024; rm -rf $O; mkdir -p $O; cd $R
for v in cf32 u8; do
  f=""; [ $v = u8 ] && f="--u8"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 bench.py --no-cpu-baseline --fs 1024000 $f > $O/bench_$v.json 2> $O/$v.err
  s=$(find $O/$v -name "*kernel_stats.csv" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --fs 1024000 $f   (fmd:: kernels only, durations in ns)"; head -1 $s; grep "fmd::" $s | cut -c1-400; echo "# bench line: $(tail -1 $O/bench_$v.json | cut -c1-2000)"; } > $O/bench_1024k_${v}_kernel_stats.csv
done
ls -la $O/*.csv
