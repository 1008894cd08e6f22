#!/bin/bash
# Development tool (GPU box): SQ counters of the stage probe's kernels, one rocprofv3 --pmc pass per counter group.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_stage
rm -rf $O && mkdir -p $O
cd $R
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- ./tools/stage_probe ${1:-4096} ${2:-1} 3 > /dev/null 2> $O/g$i.err
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fmd::", "")[:28]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
