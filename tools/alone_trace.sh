#!/bin/bash
# Development tool (GPU box; needs the development build of the library: make -C fm-radio_amd/csrc dev): rocprofv3 kernel durations of single stages of the tolerance mode run on their own (FMD_DEBUG_SKIP_STAGES).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export GPU_MAX_HW_QUEUES=8
for s in 56 41 49 31; do
  export FMD_DEBUG_SKIP_STAGES=$s
  rm -rf /tmp/alone_$s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/alone_$s -- python3 bench.py $1 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed > /dev/null 2>&1
  f=$(find /tmp/alone_$s -name "*kernel_stats.csv" | head -1)
  echo "skip=$s"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fmd::" in r["Name"] and "k_reset" not in r["Name"]:
        print("   ", r["Name"].split("(")[0][:60], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "max", round(float(r["MaxNs"]) / 1e3, 1))
PY
done
