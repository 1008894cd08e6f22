#!/bin/bash
# Per-round evidence (run on the GPU box through gpurun, from the repo root; ROUND=N names the output, default 5): the default bench line, kernel-trace statistics and a
# pipelined trace digest of the same command, and PMC passes of the un-pipelined run — counters only, one rocprofv3 run per
# counter set, no tracing domains combined with --pmc.  BENCH_ARGS=--exact for the exact mode (SFX names another configuration's output).  Outputs: gpurun_out/r<ROUND>prof<sfx>/.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
X="${BENCH_ARGS:-}"
if [ -z "${SFX+x}" ]; then SFX=""; [ -n "$X" ] && SFX="_exact"; fi      # SFX=_1024k BENCH_ARGS="--fs 1024000": another configuration
ROUND=${ROUND:-5}
O=$R/gpurun_out/r${ROUND}prof$SFX
rm -rf $O && mkdir -p $O
cd $R
export GPU_MAX_HW_QUEUES=8
Q="--no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
python3 bench.py $X > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py $X $Q > $O/bench_under_rocprof.json 2> $O/stats.err
python3 tools/trace_digest.py $O/stats 100 "pipelined bench.py $X" > $O/trace_digest.json
P="--steps 4 --warmup 1 --preroll 16 $Q --no-pipeline"
pmc() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$n -- python3 bench.py $X $P > /dev/null 2> $O/pmc_$n.err; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq_a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM
pmc sq_b SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES
pmc sq_c SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
# keep what is judged small: the per-dispatch CSVs are reduced to per-kernel averages by tools/digest_round.py, then dropped
python3 tools/digest_round.py $O $SFX
rm -rf $O/pmc_*/ $O/stats
ls -la $O
