# Development (GPU box): the default bench lines of every profiled configuration again (bench.py changed, the counters did not)
export GPU_MAX_HW_QUEUES=8
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { mkdir -p gpurun_out/r5prof$1; python3 bench.py $2 > gpurun_out/r5prof$1/bench_default.json 2> gpurun_out/r5prof$1/bench_default.err; tail -c 200 gpurun_out/r5prof$1/bench_default.json | head -c 100; echo; }
run "" ""
run _u8 "--u8"
run _1024k "--fs 1024000"
run _1024k_u8 "--fs 1024000 --u8"
run _8192 "--channels 8192"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5_evidence/bench_driver_style.json 2> gpurun_out/r5_evidence/bench_driver_style.err
