#!/bin/bash
# Development tool: static instruction mix of the tolerance-mode kernels (loops there are unrolled: the counts are per wavefront and tile,
# give or take the few loops left).  usage: tools/isa_counts.sh [pattern ...]
D=$(mktemp -d); R=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -S --cuda-device-only $R/fm-radio_amd/csrc/fmd_kernels.hip -o $D/k.s 2>/dev/null
for k in "${@:-k_front_mfmaI15HIP_vector_typeIfLj2EELi1024ELi0 k_front_mfmaI15HIP_vector_typeIhLj2EELi1024ELi0 k_extract_bp k_pll_sparse}"; do for kk in $k; do
  awk -v k="$kk" '/^_ZN3fmd/ && /:/ && index($0,k) {f=1} f{print} f&&/s_endpgm/{exit}' $D/k.s > $D/one.s
  echo "$kk: valu $(grep -cE '^\s+v_' $D/one.s) (mfma $(grep -c v_mfma $D/one.s), trans $(grep -cE 'v_(rcp|sin|cos|sqrt|rsq|exp|log)_' $D/one.s), dpp $(grep -c ' row_\| quad_perm\| wave_sh' $D/one.s), readlane $(grep -c v_readlane $D/one.s)) salu $(grep -cE '^\s+s_' $D/one.s) ds $(grep -cE '^\s+ds_' $D/one.s) vmem $(grep -cE '^\s+(global|buffer|flat)_' $D/one.s) waitcnt $(grep -c s_waitcnt $D/one.s) barrier $(grep -c s_barrier $D/one.s)"
done; done
rm -rf $D
