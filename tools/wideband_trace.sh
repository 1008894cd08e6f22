#!/bin/bash
# Development (GPU box): rocprofv3 kernel stats of bench.py --wideband (per-kernel average durations) and the bench line's own figures.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf /tmp/wb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wb -- python3 bench.py --wideband --no-cpu-baseline $1 > /tmp/wb.json 2>/dev/null
tail -1 /tmp/wb.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', round(d['ms_per_step'],4), 'x real time', round(d['realtime_factor'],1), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})"
f=$(find /tmp/wb -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print("   ", r["Name"].split("(")[0][:70], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "%", r["Percentage"])
PY
mkdir -p gpurun_out/wideband; cp "$f" gpurun_out/wideband/kernel_stats$2.csv; cp /tmp/wb.json gpurun_out/wideband/bench$2.json
