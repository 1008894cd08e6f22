#!/bin/bash
# fast-mode: per-kernel VALU instruction counts (PMC, unpipelined) and a kernel-trace timeline of the pipelined run
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_fast
rm -rf $O && mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=8
ARGS="${BENCH_ARGS:---fast-math}"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_insts -- python3 bench.py $ARGS --steps 4 --warmup 1 --preroll 8 --no-cpu-baseline --no-pipeline > /dev/null 2> $O/pmc_insts.err
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py $ARGS --steps 12 --warmup 3 --preroll 8 --no-cpu-baseline --no-kernel-times > $O/bench_trace.json 2> $O/trace.err
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_fast"
f = glob.glob(O + "/pmc_insts/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
per = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])] += float(r["Counter_Value"])
for (d, k, c), v in per.items():
    agg[k][c].append(v)
for k, cs in agg.items():
    if "fmd::" in k:
        print(k, {c: round(sum(v[-4:]) / len(v[-4:]) / 1e6, 2) for c, v in cs.items()}, "(M per launch)")
t = glob.glob(O + "/trace/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void fmd::", "")[:28]) for r in csv.DictReader(open(t)) if "fmd::" in r["Kernel_Name"]]
rows.sort()
t0 = rows[-40][0]
print("timeline of the last launches (us since first shown): start end dur name")
for s, e, n in rows[-40:]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}")
PY
cat $O/bench_trace.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('under trace:', round(d['value']), d['ms_per_step'])"
