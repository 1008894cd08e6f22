#!/bin/bash
# Development tool (GPU box): kernel trace of the pipelined library; prints per-kernel duration and start-to-start interval.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gaptrace; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/host_submit_probe.py ${1:-4096} 60 > $O/run.log 2>&1
python3 - <<PY
import csv, glob, collections, statistics as st
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fmd::" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "").split("<")[0]
    by[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in by.items():
    v.sort(); v = v[-120:]
    dur = [e - s for s, e in v]
    iv = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
    gap = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print("%-16s n=%d dur med %.1f us  start-to-start med %.1f us  gap(end->next start) med %.1f us min %.1f" % (k, len(v), st.median(dur) / 1e3, st.median(iv) / 1e3, st.median(gap) / 1e3, min(gap) / 1e3))
PY
tail -1 $O/run.log
