#!/bin/bash
# Round 6, everything profiles/round6/ is made from that depends on the library's sources, in one gpurun call (repo root): the GPU tests, the
# default line's counters / traces / stages alone / measurement table (tools/collect_evidence.sh) and the counters of the other bench
# configurations (tools/collect_round.sh).  Then, here: python tools/digest_round.py --install 6 (copies the summaries, re-stamps profiles/hbm_traffic.json).
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4
ROUND=6 bash tools/collect_evidence.sh
SFX=_u8 BENCH_ARGS="--u8" ROUND=6 bash tools/collect_round.sh > gpurun_out/collect_u8.log 2>&1
SFX=_1024k BENCH_ARGS="--fs 1024000" ROUND=6 bash tools/collect_round.sh > gpurun_out/collect_1024k.log 2>&1
SFX=_1024k_u8 BENCH_ARGS="--fs 1024000 --u8" ROUND=6 bash tools/collect_round.sh > gpurun_out/collect_1024k_u8.log 2>&1
SFX=_8192 BENCH_ARGS="--channels 8192" ROUND=6 bash tools/collect_round.sh > gpurun_out/collect_8192.log 2>&1
BENCH_ARGS="--exact" ROUND=6 bash tools/collect_round.sh > gpurun_out/collect_exact.log 2>&1
ls gpurun_out/r6_evidence gpurun_out/r6prof*
