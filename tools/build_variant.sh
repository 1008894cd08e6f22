#!/bin/bash
# Development: build a variant of the library (with the dev hooks; NODEV=1: without, i.e. the shipping flags) into tools/ab/<name>.so.  usage: tools/build_variant.sh name "-DFOO=1 -DBAR=2"
set -e
cd "$(dirname "$0")/../fm-radio_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I../../include -Wall -Wno-unused-function ${NODEV:+-UFMD_DEV_HOOKS} $2"
[ -z "$NODEV" ] && F="$F -DFMD_DEV_HOOKS"
T=/tmp/variant_$1; mkdir -p $T; COMMON=/tmp/variant_common${NODEV:+_nodev}
hipcc --offload-arch=gfx950 $F -mllvm -amdgpu-mfma-vgpr-form -c fmd_kernels.hip -o $T/fmd_kernels.o -Rpass-analysis=kernel-resource-usage 2> $T/res.txt
# (the objects shared by every variant are rebuilt when one of their sources is newer)
[ -f $COMMON/fmd_api.o ] && [ -z "$(find fmd_api.cpp fmd_channelizer.hip fmd_design.cpp fmd_kernels.h fmd_design.h ../../include -newer $COMMON/fmd_api.o 2>/dev/null)" ] || { mkdir -p $COMMON; hipcc --offload-arch=gfx950 $F -c fmd_api.cpp -o $COMMON/fmd_api.o; hipcc --offload-arch=gfx950 $F -c fmd_channelizer.hip -o $COMMON/fmd_channelizer.o; hipcc --offload-arch=gfx950 $F -x c++ -c fmd_design.cpp -o $COMMON/fmd_design.o; }
hipcc --offload-arch=gfx950 -shared -fPIC $T/fmd_kernels.o $COMMON/fmd_api.o $COMMON/fmd_design.o $COMMON/fmd_channelizer.o -o ../../tools/ab/$1.so
grep -A12 "Function Name: .*k_front_pre_mfmaILi4E15HIP_vector_typeIfLj2EELb1" $T/res.txt | grep -E "VGPRs:|Scratch|Occupancy|LDS" | sed 's/.*remark: //' | tr '\n' ' '; echo
