# Development (GPU box): every kernel of the wideband bench's last steps, whatever its namespace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --no-cpu-baseline --wideband --no-kernel-times > /tmp/tr.json 2>/tmp/tr.err
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-60:-12]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3}  {n}")
PY
