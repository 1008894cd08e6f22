#!/usr/bin/env python3
"""Development: the last launches of a rocprofv3 --kernel-trace as a timeline (start / end in us relative to the first shown, queue, kernel).
    tools/timeline.py <dir with *kernel_trace.csv> [n launches = 24]"""
import csv, glob, sys
files = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = [r for r in csv.DictReader(open(files[-1])) if "fmd::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n - 12:-12] if len(rows) > n + 12 else rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "").split("<")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3}  {name}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
