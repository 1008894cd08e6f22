#!/bin/bash
# VALU / LDS / SALU / MFMA instruction counts per kernel (unpipelined run, counters only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/insts; rm -rf $O; mkdir -p $O; cd $R
export GPU_MAX_HW_QUEUES=8
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -- python3 bench.py ${BENCH_ARGS:-} --steps 4 --warmup 1 --preroll 16 --no-cpu-baseline --no-other-mode --no-configs --no-host-fed --no-pipeline > /dev/null 2> $O/a.err
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/insts"
fs = glob.glob(O + "/a/*/*counter_collection.csv")
if not fs: print(open(O + "/a.err").read()[-2000:])
else:
    per = collections.defaultdict(float); agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])): per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items(): agg[k][c].append(v)
    tot = 0
    for k, cs in agg.items():
        if "fmd::" in k and "reset" not in k:
            row = {c.replace("SQ_", ""): round(sum(v[-3:]) / len(v[-3:]) / 1e6, 2) for c, v in cs.items()}
            tot += row.get("INSTS_VALU", 0); print(k, row)
    print("VALU total (M wave-instructions per block):", round(tot, 1))
PY
