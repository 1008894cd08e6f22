#!/bin/bash
# Round-3 (GPU box): kernel traces of the pipelined tolerance-mode bench, whole and with stages not launched, digested.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r3_trace; mkdir -p $O; rm -f $O/digest.jsonl
tr() { # label skip-mask extra-args
  rm -rf $O/t_$1; FMD_DEBUG_SKIP_STAGES=$2 rocprofv3 --kernel-trace --output-format csv -d $O/t_$1 -- python3 bench.py --steps 60 --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed $3 > $O/run_$1.log 2>&1
  grep "\"metric\"" $O/run_$1.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', sys.argv[1], round(d['value']), round(d['ms_per_step'],4))" $1
  python3 tools/trace_digest.py $O/t_$1 60 $1 | tee -a $O/digest.jsonl
  rm -rf $O/t_$1
}
# host submit rate: nothing launched at all
FMD_DEBUG_SKIP_STAGES=63 python bench.py --no-kernel-times --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host only (no kernels): ms/step', round(d['ms_per_step'],4))"
tr all 0 ""
tr front_only 56 ""
tr front_extract 40 ""
tr no_rds 32 ""
tr c1024 0 "--channels 1024"
tr u8 0 "--u8"
