#!/bin/bash
# where do the wavefronts of each kernel spend their cycles (unpipelined run)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_wait
rm -rf $O && mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=8
ARGS="${BENCH_ARGS:---fast-math}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU --output-format csv -d $O/a -- python3 bench.py $ARGS --steps 12 --warmup 2 --preroll 8 --no-cpu-baseline --no-other-mode --no-configs ${PIPE:---no-pipeline} > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -- python3 bench.py $ARGS --steps 12 --warmup 2 --preroll 8 --no-cpu-baseline --no-other-mode --no-configs ${PIPE:---no-pipeline} > /dev/null 2> $O/b.err
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc_wait"
for sub in "ab":
    fs = glob.glob(O + f"/{sub}/*/*counter_collection.csv")
    if not fs:
        print(sub, "no output", open(O + f"/{sub}.err").read()[-1500:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][:50], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        agg[k][c].append(v)
    for k, cs in agg.items():
        if "fmd::" in k and "reset" not in k:
            print(k, {c.replace("SQ_", ""): round(sum(v[-4:]) / len(v[-4:]) / 1e6, 2) for c, v in cs.items()})
PY
