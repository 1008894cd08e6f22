# Development (round 5): each throughput kernel ALONE on its part of the chip (masked queues), and with the masks swapped
cp fm-radio_amd/csrc/libfmdemod.so /tmp/orig.so; cp tools/ab/r5base.so fm-radio_amd/csrc/libfmdemod.so
B="python bench.py --no-cpu-baseline --no-other-mode --no-configs --no-host-fed"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=(d.get("roofline") or {}).get("kernels_ms_per_step") or d.get("kernels_ms_per_step") or {}; print(round(d["ms_per_step"],4), {a: round(b,3) for a,b in k.items()})'
LO=ffffffff-ffffffff-ffffffff-0000ffff-00000000-00000000-00000000-00000000; HI=00000000-00000000-00000000-ffff0000-ffffffff-ffffffff-ffffffff-ffffffff
ALL=ffffffff-ffffffff-ffffffff-ffffffff-ffffffff-ffffffff-ffffffff-ffffffff
run() { echo -n "$1: "; X=""; case "$2 $3 $4" in *SKIP*) X="--no-kernel-times";; esac; ( export $2 $3 $4; $B $X 2>/tmp/err.txt | python -c "$P" || tail -3 /tmp/err.txt ); }
run "no masks, all stages" A=1
run "no masks, front only" FMD_DEBUG_SKIP_STAGES=56
run "no masks, extract only" FMD_DEBUG_SKIP_STAGES=41
run "F=lo112 X=hi144, all" FMD_CU_MASK_F=$LO FMD_CU_MASK_X=$HI
run "F=lo112 X=hi144, front only" FMD_CU_MASK_F=$LO FMD_CU_MASK_X=$HI FMD_DEBUG_SKIP_STAGES=56
run "F=lo112 X=hi144, extract only" FMD_CU_MASK_F=$LO FMD_CU_MASK_X=$HI FMD_DEBUG_SKIP_STAGES=41
run "F=hi144 X=lo112, all" FMD_CU_MASK_F=$HI FMD_CU_MASK_X=$LO
run "F=hi144 X=lo112, extract only" FMD_CU_MASK_F=$HI FMD_CU_MASK_X=$LO FMD_DEBUG_SKIP_STAGES=41
run "F=all X=all (two plain queues side by side), all" FMD_CU_MASK_F=$ALL FMD_CU_MASK_X=$ALL
run "F=all X=hi144, all" FMD_CU_MASK_F=$ALL FMD_CU_MASK_X=$HI
run "F=lo112 X=all, all" FMD_CU_MASK_F=$LO FMD_CU_MASK_X=$ALL
cp /tmp/orig.so fm-radio_amd/csrc/libfmdemod.so
