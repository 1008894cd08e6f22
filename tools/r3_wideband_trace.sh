#!/bin/bash
# Development tool (GPU box): start / end of every kernel over a few steps of `bench.py --wideband` (rocprofv3 --kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf /tmp/wb; rocprofv3 --kernel-trace --output-format csv -d /tmp/wb -- python3 bench.py --wideband --steps 40 --no-kernel-times $1 > /tmp/wb.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/wb/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "k_reset" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-64:-16]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "").split("<")[0][:28]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{n:28s} q{r.get('Queue_Id','?'):3s} start {s:8.1f} end {e:8.1f} dur {e - s:7.1f}")
PY
