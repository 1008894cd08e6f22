export GPU_MAX_HW_QUEUES=8
run() { FMD_DEBUG_SKIP_STAGES=$2 python bench.py $3 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'cfg': sys.argv[1], 'value': round(d['value']), 'ms': round(d['ms_per_step'],4)}))" "$1"; }
run "front only pipelined" 56 ""
run "front only no-pipeline" 56 "--no-pipeline"
run "extract only no-pipeline" 41 "--no-pipeline"
run "extract only pipelined" 41 ""
run "pll only no-pipeline" 49 "--no-pipeline"
run "pll only pipelined" 49 ""
run "rds only no-pipeline" 25 "--no-pipeline"
run "rds only pipelined" 25 ""
run "all no-pipeline" 0 "--no-pipeline"
