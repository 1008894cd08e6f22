#!/bin/bash
# Development tool (GPU box): start / end of every kernel launch over a few steps of the pipelined tolerance-mode bench (rocprofv3 --kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export GPU_MAX_HW_QUEUES=8
rm -rf /tmp/tl; FMD_DEBUG_SKIP_STAGES=${SKIP:-0} rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 60 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed $1 > /tmp/tl.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "fmd::" in r["Kernel_Name"] and "k_reset" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:-8]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "").split("<")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{n:16s} q{r.get('Queue_Id','?'):3s} start {s:8.1f} end {e:8.1f} dur {e - s:7.1f}")
PY
