#!/bin/bash
# Development (round 5): the front end's queue and the extract stage's queue on disjoint CU sets (FMD_CU_MASK_F / FMD_CU_MASK_X of the
# development build) against the shipping schedule, interleaved on one box.  usage: tools/cumask_ab.sh <variant> "K list" [bench args] [rounds]
#   family "lin": the first K of 256 mask bits to the front end; family "word": the low K/8 bits of every 32-bit word
V=${1:-r5base}; KS=${2:-"96 128 160"}; ARGS=$3; R=${4:-2}
mk() { python3 - "$1" "$2" <<'PY'
import sys
fam, k = sys.argv[1], int(sys.argv[2])
bits = [0] * 256
if fam == "lin":
    for i in range(k): bits[i] = 1
else:
    for w in range(8):
        for i in range(k // 8): bits[32 * w + i] = 1
def words(b): return "-".join("%08x" % sum(b[32 * w + i] << i for i in range(32)) for w in range(8))
print("FMD_CU_MASK_F=" + words(bits) + ",FMD_CU_MASK_X=" + words([1 - x for x in bits]))
PY
}
SPECS="$V"
for fam in lin; do for k in $KS; do SPECS="$SPECS $V:$(mk $fam $k)"; done; done
tools/ab.sh "$SPECS" $R "$ARGS"
