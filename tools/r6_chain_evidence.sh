#!/bin/bash
# Round 6: the one-launch form of a steady block (k_chain, fmd_kernels_chain.inc) as a measured A/B against the three-launch
# form, with its counters.  Needs tools/ab/chain.so (development build: tools/build_variant.sh chain "") and tools/ab/cprobe.so
# (tools/build_variant.sh cprobe "-DFMD_C_PROBE").  Output: gpurun_out/r6_chain/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export GPU_MAX_HW_QUEUES=8
O=$R/gpurun_out/r6_chain; rm -rf $O; mkdir -p $O gpurun_out/ab
L=fm-radio_amd/csrc/libfmdemod.so; cp $L /tmp/r6_orig.so
Q="--no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
# 1. bench lines, interleaved: three launches / one launch, with and without the RDS stage beside them
rm -f gpurun_out/ab/table.txt
tools/ab.sh "chain chain:FMD_CHAIN=1" 3 ""
tools/ab.sh "chain chain:FMD_CHAIN=1" 2 "--channels 2048"
tools/ab.sh "chain chain:FMD_CHAIN=1" 2 "--channels 8192"
cp gpurun_out/ab/table.txt $O/bench_ab.txt
cp tools/ab/chain.so $L
for c in 2048 4096 8192; do for e in "" "FMD_CHAIN=1"; do
  ( export FMD_DEBUG_SKIP_STAGES=32 $e; python3 bench.py --channels $c $Q --no-kernel-times 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('without the RDS stage:', sys.argv[1:], round(d['value']), 'MSa/s', round(d['ms_per_step'],4), 'ms')" $c $e ) | tee -a $O/bench_without_rds.txt
done; done
# 2. where the wavefronts' cycles go, and how many workgroups are resident at once
cp tools/ab/cprobe.so $L
for n in 2048 4096; do echo "== $n stations, with the RDS stage"; python3 tools/dbg/chain_probe.py $n 2>&1 | grep -v amdgpu.ids; echo "== $n stations, without"; FMD_DEBUG_SKIP_STAGES=32 python3 tools/dbg/chain_probe.py $n 2>&1 | grep -v amdgpu.ids; done > $O/chain_probe.txt
python3 tools/dbg/chain_check.py 11 10 2>&1 | grep -v amdgpu.ids > $O/chain_check.txt
# 3. counters of the pipelined run with k_chain (counters only, one rocprofv3 run per set)
cp tools/ab/chain.so $L
cd /tmp && export TMPDIR=/tmp
P="--steps 6 --warmup 2 --preroll 16 $Q"
pmc() { n=$1; shift; FMD_CHAIN=1 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$n -- python3 $R/bench.py $P > /dev/null 2> $O/pmc_$n.err; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq_a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM
pmc sq_b SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES
cd $R
python3 - <<'PY' > $O/chain_counters.json
import json, pathlib, sys
sys.path.insert(0, "tools")
import digest_round as D
o = pathlib.Path("gpurun_out/r6_chain")
c = {}
for sub in ("fetch", "write", "sq_a", "sq_b"):
    for k, cs in D.pmc_dir(o / f"pmc_{sub}").items():
        c.setdefault(k, {}).update(cs)
for k, cs in c.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        cs["hbm_bytes_corrected"] = (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0      # KiB; gfx950 FETCH_SIZE counts half of a 16 B/lane stream (MI355X_MICROARCH.md)
print(json.dumps({"what": "per-kernel averages per launch, 4096 stations x 16384 samples @ 256 kSa/s cf32, tolerance mode with fmd_debug_set_chain (FMD_CHAIN=1), pipelined", "kernels": c}, indent=1))
PY
rm -rf $O/pmc_*/
cp /tmp/r6_orig.so $L
ls -la $O
