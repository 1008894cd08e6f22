// Host-side coefficient design for the batched FM demodulator (product code, runs on the CPU at
// create / control-change time only; the reference does the same inside its constructor and
// UpdateFilters(), reference src/fm_demod/broadcast_fm_demod.cpp:127-291,330-389).
//
// Designs follow reference src/dsp/filter_designer.cpp (windowed-sinc LPF :84-107 with the Hamming
// window of window_functions.h:10-13, bilinear 1-pole LPF with pre-warp :46-64,158-200, pole-placement
// peak filter :260-310, Hilbert taps :369-384).  The arithmetic is spelled out operation by operation
// (compiled with -ffp-contract=off) in the order the reference's -ffast-math build evaluates it, so the
// taps are the same floats; the one deliberate difference is the peak filter's normalising gain, which
// the reference build approximates with rsqrtss and this code computes with IEEE sqrt and divide.
#include "fmd_design.h"

#include <math.h>
#include <string.h>

namespace fmd {

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static const float kPi = 3.14159274101257324219f;          // (float)M_PI
static const float kTwoPi = 6.28318548202514648438f;
static const float kHalfPi = 1.57079637050628662109f;
static const float kInvPi = 0.318309873342514038086f;      // 0x3ea2f983
static const float kTwoOverPi = 0.636619746685028076172f;  // 0x3f22f983
static const float kInv4Pi = 0.0795774683356285095215f;    // 0x3da2f983
static const float kMegaOverPi = 318309.875f;              // 0x489b6cbc

static inline float clampf(float x, float lo, float hi) {
    float y = (x > lo) ? x : lo;
    return (y < hi) ? y : hi;
}

void design_fir_lpf(float* b, int n, float k) {
    const float M = (float)(n - 1);
    const float half_m = M * 0.5f;
    const float step = kTwoPi / M;
    for (int i = 0; i < n; i++) {
        const float fi = (float)i;
        const float t1 = fi - half_m;
        const float t0 = fi * step;
        const float w = fmaf(-cosf(t0), 0.46164f, 0.53836f);
        const float xk = t1 * k;
        float sinc = 1.0f;
        if (!(fabsf(xk) <= 1e-6f)) sinc = (sinf(xk * kPi) * kInvPi) / xk;
        b[(n - 1) - i] = (w * k) * sinc;
    }
}

void design_hilbert(float* b, int n) {
    const int M = (n - 1) / 2;
    for (int i = 0; i < n; i++) {
        const int nn = i - M;
        b[(n - 1) - i] = ((nn % 2) == 0) ? 0.0f : kTwoOverPi / (float)nn;
    }
}

void design_iir_lpf(float* b, float* a, float k) {
    const float t = tanf(k * kHalfPi);
    const float two_a = 1.0f / t;
    const float B0 = two_a + 1.0f;
    const float b0 = 1.0f / B0;
    const float B1 = 1.0f - two_a;
    b[0] = b0; b[1] = b0;
    a[0] = -(B1 / B0); a[1] = 1.0f;
}

void design_iir_peak(float* b, float* a, float k, float r) {
    const float wn = k * kPi;
    const float s_wn = sinf(wn), c_wn = cosf(wn);
    const float s_z1 = sinf(-wn);
    const float d_im1 = fmaf(-s_z1, r, s_wn);
    const float d_im0 = fmaf(-r, s_wn, s_wn);
    const float d_re = fmaf(-r, c_wn, c_wn);
    const float D_im = (d_im0 + d_im1) * d_re;
    const float D_re = fmaf(d_re, d_re, -(d_im0 * d_im1));
    const float nrm = fmaf(D_re, D_re, D_im * D_im);
    const float h_im = (-D_im) / nrm;
    const float h_re = D_re / nrm;
    const float mag2 = fmaf(h_re, h_re, h_im * h_im);
    const float K = 1.0f / sqrtf(mag2);
    b[0] = K; b[1] = 0.0f; b[2] = 0.0f;
    a[0] = -(r * r); a[1] = c_wn * (r + r); a[2] = 1.0f;
}

static float cutoff_k(float fc, float fs) { return clampf(fc / (fs / 2.0f), 0.01f, 0.99f); }

void design_controls(fmd_coeffs* k, const fmd_controls* c) {
    float kd = kMegaOverPi / ((float)c->deemphasis_tus * 128000.0f);
    kd = clampf(kd, 0.01f, 0.99f);
    design_iir_lpf(k->deemph_b, k->deemph_a, kd);
    design_fir_lpf(k->b_lpr, 128, cutoff_k((float)c->lpr_cutoff_hz, 128000.0f));
    design_fir_lpf(k->b_lmr, 128, cutoff_k((float)c->lmr_cutoff_hz, 128000.0f));
}

void design_all(fmd_coeffs* k, int fs_baseband, const fmd_controls* c) {
    memset(k, 0, sizeof(*k));
    k->fs_baseband = fs_baseband;
    k->m_fm_in = fs_baseband / 256000;
    if (k->m_fm_in > 1) design_fir_lpf(k->b_fm_in, 64, (128000.0f / ((float)fs_baseband / 2.0f)) * 0.95f);
    design_fir_lpf(k->b_fm_out, 64, (64000.0f / (256000.0f / 2.0f)) * 0.95f);
    design_hilbert(k->b_hilbert, 65);
    design_iir_peak(k->pilot_b, k->pilot_a, 19000.0f / (128000.0f / 2.0f), 0.9999f);
    design_iir_lpf(k->pll_lpf_b, k->pll_lpf_a, 100.0f / (128000.0f / 2.0f));
    design_controls(k, c);
    design_fir_lpf(k->b_rds, 128, 2000.0f / (128000.0f / 2.0f));
    design_iir_lpf(k->ted_lpf_b, k->ted_lpf_a, 1500.0f / (16000.0f / 2.0f));
    design_iir_lpf(k->bpsk_lpf_b, k->bpsk_lpf_a, 10.0f / (16000.0f / 2.0f));
    k->fm_gain = kInv4Pi / (75e3f / 256000.0f);
    (void)u2f;
}

}  // namespace fmd
