// Host-side coefficient design (see fmd_design.cpp).
#pragma once

#include "fmdemod.h"

namespace fmd {

void design_fir_lpf(float* b, int n, float k);
void design_hilbert(float* b, int n);
void design_iir_lpf(float* b, float* a, float k);
void design_iir_peak(float* b, float* a, float k, float r);
// controls-dependent subset: de-emphasis, L+R and L-R low-pass (reference UpdateFilters())
void design_controls(fmd_coeffs* k, const fmd_controls* c);
void design_all(fmd_coeffs* k, int fs_baseband, const fmd_controls* c);

}  // namespace fmd
