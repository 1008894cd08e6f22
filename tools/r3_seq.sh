#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r3_seq; rm -rf $O; mkdir -p $O
for m in 56 0; do
FMD_DEBUG_SKIP_STAGES=$m python3 tools/r3_seq.py | cut -c1-1500
FMD_DEBUG_SKIP_STAGES=$m rocprofv3 --kernel-trace --output-format csv -d $O/t$m -- python3 tools/r3_seq.py > $O/run$m.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/t$m/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "fmd::" in r["Kernel_Name"] and "k_reset" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-100:]
t0 = int(rows[0]["Start_Timestamp"])
print("skip=$m: kernel start/end (us since first), queue")
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "").split("<")[0]
    print("%-16s q%s %8.1f %8.1f" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
PY
done
