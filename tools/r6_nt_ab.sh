#!/bin/bash
# Round 6: the HBM roof re-measured, then non-temporal IQ loads / audio stores as an A/B on one box.
# Variants: tools/build_variant.sh ntXY "-DFMD_NT_IN=X -DFMD_NT_OUT=Y".  Output: gpurun_out/r6_nt/.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r6_nt; mkdir -p $O
tools/hbm_roof_probe 2>&1 | tee $O/hbm_roof_probe.txt
rm -f gpurun_out/ab/table.txt
tools/ab.sh "nt00 nt11 nt10 nt01" 3 ""
tools/ab.sh "nt00 nt11 nt10 nt01" 3 "--fs 1024000"
tools/ab.sh "nt00 nt11" 2 "--u8"
tools/ab.sh "nt00 nt11" 2 "--fs 1024000 --u8"
tools/ab.sh "nt00 nt11" 2 "--channels 8192"
cp gpurun_out/ab/table.txt $O/nt_ab.txt
