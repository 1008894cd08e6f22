#!/bin/bash
# Development tool (GPU box): the driver's 20-step command under rocprofv3 --kernel-trace: the kernels at the start and at the end of the
# timed region (fill and drain of the pipeline).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf /tmp/fd; rocprofv3 --kernel-trace --output-format csv -d /tmp/fd -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-mode --no-configs --no-host-fed $1 > /tmp/fd.log 2>&1
tail -1 /tmp/fd.log | cut -c1-200
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/fd/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "fmd::" in r["Kernel_Name"] and "k_reset" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fronts = [i for i, r in enumerate(rows) if "k_front_mfma" in r["Kernel_Name"]]
i0 = fronts[-20]
reg = rows[i0:]
t0 = int(reg[0]["Start_Timestamp"])
def show(rs):
    for r in rs:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "").split("<")[0]
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{n:16s} q{r.get('Queue_Id','?'):3s} start {s:8.1f} end {e:8.1f} dur {e - s:7.1f}")
prev = rows[max(0, i0 - 6):i0]
print("-- before the region"); show(prev)
print("-- start"); show(reg[:14]); print("-- end"); show(reg[-12:])
print("region span us", (max(int(r["End_Timestamp"]) for r in reg) - t0) / 1e3)
PY
