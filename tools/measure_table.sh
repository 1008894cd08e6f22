#!/bin/bash
# Development tool (GPU box): the configurations of DESIGN.md's measurement table, one JSON line each (gpurun_out/table.jsonl).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out/table.jsonl; : > $O
run() { echo "# $*" >> $O; python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $O; }
run
run --no-kernel-times
run --no-pipeline
run --u8
run --channels 1024
run --channels 2048
run --fs 1024000
run --fs 1024000 --u8
run --fs 2048000 --channels 2048
run --channels 8192
run --channels 16384 --steps 40
run --channels 65536 --steps 10 --preroll 8
run --channels 1 --fs 2048000
run --wideband
python3 - <<PY
import json
lines = open("$O").read().splitlines()
for i in range(0, len(lines), 2):
    d = json.loads(lines[i + 1])
    r = d.get("roofline") or {}
    print("%-45s %9.0f MSa/s %7.3f ms  frac %s  %s" % (lines[i], d["value"], d["ms_per_step"], ("%.3f" % r["frac"]) if r else "-", {k: round(v, 2) for k, v in (r.get("kernels_ms_per_step") or {}).items()}))
PY
