timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4
ROUND=5 bash tools/collect_evidence.sh
SFX=_u8 BENCH_ARGS="--u8" ROUND=5 bash tools/collect_round.sh > gpurun_out/collect_u8.log 2>&1
SFX=_1024k BENCH_ARGS="--fs 1024000" ROUND=5 bash tools/collect_round.sh > gpurun_out/collect_1024k.log 2>&1
SFX=_1024k_u8 BENCH_ARGS="--fs 1024000 --u8" ROUND=5 bash tools/collect_round.sh > gpurun_out/collect_1024k_u8.log 2>&1
SFX=_8192 BENCH_ARGS="--channels 8192" ROUND=5 bash tools/collect_round.sh > gpurun_out/collect_8192.log 2>&1
python bench.py --wideband --no-cpu-baseline > gpurun_out/bench_wideband_r5.json 2>/dev/null
ls gpurun_out/r5_evidence gpurun_out/r5prof*
