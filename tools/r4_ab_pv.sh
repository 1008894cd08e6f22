#!/bin/bash
# Development A/B (GPU box): variants of the front end's pilot column sums (tools/ab/pv*.so): the front end alone, and the bench line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
L=fm-radio_amd/csrc/libfmdemod.so; cp $L /tmp/orig.so
O=gpurun_out/r4_ab_pv; mkdir -p $O; rm -f $O/table.txt
export GPU_MAX_HW_QUEUES=8
Q="--no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
for rep in 1 2; do for v in $1; do
  cp tools/ab/$v.so $L
  export FMD_DEBUG_SKIP_STAGES=56
  rm -rf /tmp/alone_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/alone_$v -- python3 bench.py --no-kernel-times $Q > /dev/null 2>&1
  f=$(find /tmp/alone_$v -name "*kernel_stats.csv" | head -1)
  a=$(python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_front_mfma" in r["Name"]: print(round(float(r["AverageNs"]) / 1e3, 1))
PY
)
  unset FMD_DEBUG_SKIP_STAGES
  b=$(FMD_NO_FUSED_PLL=1 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))")
  c=$(python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))")
  echo "$v front alone us: $a | unfused: $b | fused: $c" | tee -a $O/table.txt
done; done
cp /tmp/orig.so $L
