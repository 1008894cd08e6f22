#!/bin/bash
# Development A/B on one GPU box (through gpurun, repo root): bench lines of library variants, interleaved so that box and clock drift hit all alike.
#   tools/ab.sh "v1 v2:FMD_X=1 ..." [rounds] ["bench args"] [alone]
# A variant is tools/ab/<name>.so — built by tools/build_variant.sh name "-DFLAGS" (development hooks on), or a copy of any build of the
# library — optionally with environment switches of the development build behind a colon (comma separated).  Bench args may hold --wideband.
# With a fourth argument the front end's kernel is also timed on its own (FMD_DEBUG_SKIP_STAGES=56, development builds only).
L=fm-radio_amd/csrc/libfmdemod.so; cp $L /tmp/ab_orig.so
O=gpurun_out/ab; mkdir -p $O
B="python bench.py --no-cpu-baseline $3"
case "$3" in *--wideband*) ;; *) B="$B --no-other-mode --no-configs --no-host-fed";; esac
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernels_ms_per_step") or d["roofline"]["kernels_ms_per_step"]; print(round(d["value"]), round(d["ms_per_step"],4), ("x%.0f" % d["realtime_factor"]) if "realtime_factor" in d else "", {a: round(b,3) for a,b in k.items()})'
for r in $(seq 1 ${2:-2}); do for spec in $1; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=${spec#*:}
  cp tools/ab/$name.so $L
  a=$( ( for e in ${envs//,/ }; do export $e; done; $B 2>/dev/null ) | python -c "$P" )
  b=""
  [ -n "$4" ] && b=$( ( for e in ${envs//,/ }; do export $e; done; FMD_DEBUG_SKIP_STAGES=56 $B --no-kernel-times 2>/dev/null ) | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("front end alone", round(d["ms_per_step"],4))' )
  echo "$spec [$3] $a $b" | tee -a $O/table.txt
done; done
cp /tmp/ab_orig.so $L
