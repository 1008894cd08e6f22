/* TEST INFRASTRUCTURE — CPU restatement of the reference FM demodulation path (see fm_oracle.h).
 *
 * Build: gcc -O2 -std=c11 -mavx2 -mfma -ffp-contract=off  (no fast-math: every fused
 * multiply-add below is an explicit fmaf(), every other operation rounds on its own, so
 * the arithmetic is pinned by this source rather than by compiler flags).
 *
 * Operation orders were read from the disassembly of the reference's `gcc` preset build
 * (g++ 11.4 -O2 -ffast-math, AVX2+FMA; reference CMakePresets.json:32-40) and are checked
 * bit-for-bit against that build by tests/test_oracle_vs_ref.py.
 */
#define _GNU_SOURCE
#include "fm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

/* float constants exactly as the reference build materialises them */
static inline float f32_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#define PI_F          f32_bits(0x40490fdbu) /* (float)M_PI  */
#define TWO_PI_F      f32_bits(0x40c90fdbu) /* 2*pi         */
#define HALF_PI_F     f32_bits(0x3fc90fdbu) /* pi/2         */
#define INV_PI_F      f32_bits(0x3ea2f983u) /* 1/pi         */
#define TWO_OVER_PI_F f32_bits(0x3f22f983u) /* 2/pi         */
#define INV_4PI_F     f32_bits(0x3da2f983u) /* 1/(4*pi)     */
#define MEGA_OVER_PI_F f32_bits(0x489b6cbcu) /* 1e6/pi       */
#define PRED_HALF_F   f32_bits(0x3effffffu) /* nextafter(0.5, 0) */

/* std::round as the reference build inlines it (-ffast-math): trunc(x + copysign(pred(0.5), x)). */
static inline float round_half_away(float x) { return truncf(x + copysignf(PRED_HALF_F, x)); }

/* reference src/dsp/clamp.h:3-8 (compiled to vmaxss/vminss) */
static inline float clampf(float x, float lo, float hi) {
    float y = (x > lo) ? x : lo;
    y = (y < hi) ? y : hi;
    return y;
}

/* ------------------------------------------------------------------------------------------
 * chebyshev sine — reference src/dsp/simd/chebyshev_sine.h:13-41 (scalar) and :78-104 (AVX).
 * The two compiled forms associate the final product differently.
 * ---------------------------------------------------------------------------------------- */
static const float CHEB_A[6] = { -25.13274193f, 64.83583069f, -67.07687378f, 38.50016403f, -14.07150173f, 3.20396066f };

static inline float cheb_poly(float z) {
    float p = fmaf(CHEB_A[5], z, CHEB_A[4]);
    p = fmaf(p, z, CHEB_A[3]);
    p = fmaf(p, z, CHEB_A[2]);
    p = fmaf(p, z, CHEB_A[1]);
    p = fmaf(p, z, CHEB_A[0]);
    return p;
}
/* scalar call sites (pilot PLL loop, BPSK loop): ((z-0.25)*x)*g(z) */
float fmo_chebyshev_sine(float x) {
    const float z = x * x;
    const float g = cheb_poly(z);
    return ((z - 0.25f) * x) * g;
}
/* AVX call site (apply_harmonic_pll_avx): (x*g(z))*(z + -0.25) */
static inline float cheb_sine_avx(float x) {
    const float z = x * x;
    const float g = cheb_poly(z);
    return (x * g) * (z + -0.25f);
}

/* ------------------------------------------------------------------------------------------
 * dot products — reference src/dsp/simd/f32_cum_mul.cpp:52-78, c32_f32_cum_mul.cpp:70-111,
 * horizontal sums x86/f32_cum_sum.h:10-40, x86/c32_cum_sum.h:10-39.
 * ---------------------------------------------------------------------------------------- */
float fmo_dot_f32(const float* x, const float* b, int n) {
    float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const int nv = (n / 8) * 8;
    for (int i = 0; i < nv; i += 8)
        for (int j = 0; j < 8; j++) acc[j] = fmaf(x[i + j], b[i + j], acc[j]);
    const float a0 = acc[0] + acc[4], a1 = acc[1] + acc[5], a2 = acc[2] + acc[6], a3 = acc[3] + acc[7];
    const float s0 = a0 + a2, s1 = a1 + a3;
    float y = s0 + s1;
    if (n > nv) {
        float t = 0.0f;
        for (int i = nv; i < n; i++) t = fmaf(x[i], b[i], t);
        y = y + t;
    }
    return y;
}

fmo_cf32 fmo_dot_c32(const fmo_cf32* x, const float* b, int n) {
    float ar[4] = { 0, 0, 0, 0 }, ai[4] = { 0, 0, 0, 0 };
    const int nv = (n / 4) * 4;
    for (int i = 0; i < nv; i += 4)
        for (int j = 0; j < 4; j++) {
            ar[j] = fmaf(x[i + j].re, b[i + j], ar[j]);
            ai[j] = fmaf(x[i + j].im, b[i + j], ai[j]);
        }
    fmo_cf32 y;
    y.re = (ar[0] + ar[2]) + (ar[1] + ar[3]);
    y.im = (ai[0] + ai[2]) + (ai[1] + ai[3]);
    if (n > nv) {
        float tr = 0.0f, ti = 0.0f;
        for (int i = nv; i < n; i++) { tr = fmaf(b[i], x[i].re, tr); ti = fmaf(b[i], x[i].im, ti); }
        y.re = y.re + tr;
        y.im = y.im + ti;
    }
    return y;
}

/* ------------------------------------------------------------------------------------------
 * decimating FIR with carried history — reference src/dsp/polyphase_filter.h:41-64:
 * output i is the dot product over the NN most recent inputs ending at stream index M(i+1)-1.
 * hist holds the last NN inputs of the stream (zeros at start).
 * ---------------------------------------------------------------------------------------- */
void fmo_decim_c32(fmo_cf32* hist, const float* b, int nn, int m, const fmo_cf32* x, fmo_cf32* y, int n_out) {
    const size_t n_in = (size_t)m * (size_t)n_out;
    fmo_cf32* cat = (fmo_cf32*)malloc(sizeof(fmo_cf32) * (nn + n_in));
    memcpy(cat, hist, sizeof(fmo_cf32) * nn);
    memcpy(cat + nn, x, sizeof(fmo_cf32) * n_in);
    for (int i = 0; i < n_out; i++) y[i] = fmo_dot_c32(cat + (size_t)m * (i + 1), b, nn);
    memcpy(hist, cat + n_in, sizeof(fmo_cf32) * nn);
    free(cat);
}

void fmo_decim_f32(float* hist, const float* b, int nn, int m, const float* x, float* y, int n_out) {
    const size_t n_in = (size_t)m * (size_t)n_out;
    float* cat = (float*)malloc(sizeof(float) * (nn + n_in));
    memcpy(cat, hist, sizeof(float) * nn);
    memcpy(cat + nn, x, sizeof(float) * n_in);
    for (int i = 0; i < n_out; i++) y[i] = fmo_dot_f32(cat + (size_t)m * (i + 1), b, nn);
    memcpy(hist, cat + n_in, sizeof(float) * nn);
    free(cat);
}

/* Hilbert FIR — reference src/dsp/hilbert_fir_filter.h:26-46 (K = 65): y[i] = { s[i-32], dot(s[i-64..i], b) } */
void fmo_hilbert(float* hist65, const float* b65, const float* x, fmo_cf32* y, int n) {
    const int K = 65, M = 32;
    float* cat = (float*)malloc(sizeof(float) * (K + (size_t)n));
    memcpy(cat, hist65, sizeof(float) * K);
    memcpy(cat + K, x, sizeof(float) * (size_t)n);
    for (int i = 0; i < n; i++) {
        y[i].im = fmo_dot_f32(cat + i + 1, b65, K);
        y[i].re = cat[i + 1 + M];
    }
    memcpy(hist65, cat + n, sizeof(float) * K);
    free(cat);
}

/* ------------------------------------------------------------------------------------------
 * IIR, direct form I — reference src/dsp/iir_filter.h:40-69.  xn[k-1] newest x, yn[k-2] = y[n-1],
 * yn[k-1] stays 0.  Compiled forms:  float   t = fma(xn[i], b[i], yn[i]*a[i]);  y += t
 *                                    complex t = fma(a[i], yn[i], b[i]*xn[i]);  y += t  (per rail)
 * ---------------------------------------------------------------------------------------- */
static inline float iir_f32_step(const float* b, const float* a, int k, float* xn, float* yn, float x) {
    for (int i = 0; i < k - 1; i++) xn[i] = xn[i + 1];
    xn[k - 1] = x;
    float y = 0.0f;
    for (int i = 0; i < k; i++) {
        const float t = fmaf(xn[i], b[i], yn[i] * a[i]);
        y = y + t;
    }
    for (int i = 0; i < k - 2; i++) yn[i] = yn[i + 1];
    yn[k - 2] = y;
    return y;
}

void fmo_iir_f32(const float* b, const float* a, int k, float* xn, float* yn, const float* x, float* y, int n) {
    for (int i = 0; i < n; i++) y[i] = iir_f32_step(b, a, k, xn, yn, x[i]);
}

void fmo_iir_c32(const float* b, const float* a, int k, fmo_cf32* xn, fmo_cf32* yn, const fmo_cf32* x, fmo_cf32* y, int n) {
    for (int s = 0; s < n; s++) {
        for (int i = 0; i < k - 1; i++) xn[i] = xn[i + 1];
        xn[k - 1] = x[s];
        float yr = 0.0f, yi = 0.0f;
        for (int i = 0; i < k; i++) {
            const float tr = fmaf(a[i], yn[i].re, b[i] * xn[i].re);
            const float ti = fmaf(a[i], yn[i].im, b[i] * xn[i].im);
            yr = yr + tr;
            yi = yi + ti;
        }
        for (int i = 0; i < k - 2; i++) yn[i] = yn[i + 1];
        yn[k - 2].re = yr; yn[k - 2].im = yi;
        y[s].re = yr; y[s].im = yi;
    }
}

/* AGC — reference src/dsp/agc.h:12-30; compiled: gain_target = sqrt((target/sum)*N) */
float fmo_agc(float* gain, float target_power, float beta, const fmo_cf32* x, fmo_cf32* y, int n) {
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        const float t = fmaf(x[i].re, x[i].re, x[i].im * x[i].im);
        sum = sum + t;
    }
    const float target_gain = sqrtf((target_power / sum) * (float)n);
    const float g = fmaf(target_gain - *gain, beta, *gain);
    *gain = g;
    for (int i = 0; i < n; i++) { y[i].re = g * x[i].re; y[i].im = g * x[i].im; }
    return g;
}

/* the host libm atan2f over arrays (what std::atan2 resolves to in the reference build); used by tests to pin the device math */
void fmo_atan2f_array(const float* y, const float* x, float* out, long n) {
    for (long i = 0; i < n; i++) out[i] = atan2f(y[i], x[i]);
}

/* FM discriminator — reference src/fm_demod/fm_demod.cpp:30-45 */
void fmo_discriminator(float* prev_theta, float gain, const fmo_cf32* x, float* y, int n) {
    float prev = *prev_theta;
    for (int i = 0; i < n; i++) {
        const float th = atan2f(x[i].im, x[i].re);
        float d = th - prev;
        if (d >= PI_F) d = d - TWO_PI_F;
        else if (d <= -PI_F) d = d + TWO_PI_F;
        y[i] = d * gain;
        prev = th;
    }
    *prev_theta = prev;
}

/* harmonic mixer — reference src/dsp/simd/apply_harmonic_pll.cpp:88-139 (AVX, 4 samples per step,
 * round-half-even) with the scalar tail :11-24 (round-half-away) for n % 4 samples. */
void fmo_harmonic_mix(const float* dt, const fmo_cf32* x, fmo_cf32* y, int n, float harmonic, float offset) {
    const int nv = (n / 4) * 4;
    const float off_cos = offset + 0.25f;
    for (int i = 0; i < nv; i++) {
        float s = fmaf(dt[i], harmonic, offset);
        float c = fmaf(dt[i], harmonic, off_cos);
        s = s - rintf(s);
        c = c - rintf(c);
        const float pc = cheb_sine_avx(c), ps = cheb_sine_avx(s);
        const float p = x[i].re, q = x[i].im;
        y[i].re = fmaf(pc, p, -(q * ps));
        y[i].im = fmaf(pc, q, p * ps);
    }
    for (int i = nv; i < n; i++) {
        float s = fmaf(dt[i], harmonic, offset);
        float c = s + 0.25f;
        s = s - round_half_away(s);
        c = c - round_half_away(c);
        const float pc = fmo_chebyshev_sine(c), ps = fmo_chebyshev_sine(s);
        const float p = x[i].re, q = x[i].im;
        y[i].re = fmaf(p, pc, -(q * ps));
        y[i].im = fmaf(p, ps, q * pc);
    }
}

/* ------------------------------------------------------------------------------------------
 * filter designer — reference src/dsp/filter_designer.cpp (as compiled with -ffast-math)
 * ---------------------------------------------------------------------------------------- */
/* :84-107 with window_hamming (window_functions.h:10-13); taps stored reversed */
void fmo_design_fir_lpf(float* b, int n, float k) {
    const float M = (float)(n - 1);
    const float half_m = M * 0.5f;
    const float step = TWO_PI_F / M;
    for (int i = 0; i < n; i++) {
        const float fi = (float)i;
        const float t1 = fi - half_m;
        const float t0 = fi * step;
        const float w = fmaf(-cosf(t0), 0.46164f, 0.53836f);
        const float xk = t1 * k;
        float sinc = 1.0f;
        if (!(fabsf(xk) <= 1e-6f)) sinc = (sinf(xk * PI_F) * INV_PI_F) / xk;
        b[(n - 1) - i] = (w * k) * sinc;
    }
}

/* :369-384 */
void fmo_design_hilbert(float* b, int n) {
    const int M = (n - 1) / 2;
    for (int i = 0; i < n; i++) {
        const int nn = i - M;
        b[(n - 1) - i] = ((nn % 2) == 0) ? 0.0f : TWO_OVER_PI_F / (float)nn;
    }
}

/* :158-200 (+ prewarp :46-64); arrays newest-last: b = [b0, b0], a = [-a0, 1] */
void fmo_design_iir_lpf(float* b, float* a, float k) {
    const float t = tanf(k * HALF_PI_F);
    const float two_a = 1.0f / t;
    const float B0 = two_a + 1.0f;
    const float b0 = 1.0f / B0;
    const float B1 = 1.0f - two_a;
    const float a0 = B1 / B0;
    b[0] = b0; b[1] = b0;
    a[0] = -a0; a[1] = 1.0f;
}

/* :260-310; arrays newest-last: b = [K, 0, 0], a = [-r^2, 2r*cos(pi k), 1] */
void fmo_design_iir_peak(float* b, float* a, float k, float r, int rsqrt_mode) {
    const float wn = k * PI_F;
    float s_wn, c_wn;
    sincosf(wn, &s_wn, &c_wn);
    const float two_r = r + r;
    const float r2 = r * r;
    /* H(z) at z = exp(j*pi*k): 1/((z - r z0)(z - r z1)), z0,1 = exp(+-j*pi*k) */
    float s_z0, c_z0;
    sincosf(PI_F * k, &s_z0, &c_z0);
    const float s_z1 = sinf(-(PI_F * k));
    const float d_im1 = fmaf(-s_z1, r, s_wn); /* z.im - r*z1.im */
    const float d_im0 = fmaf(-r, s_z0, s_wn); /* z.im - r*z0.im */
    const float d_re = fmaf(-r, c_z0, c_wn);  /* z.re - r*z0.re */
    const float D_im = (d_im0 + d_im1) * d_re;
    const float D_re = fmaf(d_re, d_re, -(d_im0 * d_im1));
    const float nrm = fmaf(D_re, D_re, D_im * D_im);
    const float h_im = (-D_im) / nrm;
    const float h_re = D_re / nrm;
    const float mag2 = fmaf(h_re, h_re, h_im * h_im);
    float K;
    if (rsqrt_mode) {
        /* 1/sqrt via rsqrtss + one Newton-Raphson step, as -ffast-math emits it */
        const float x0 = _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(mag2)));
        const float e = fmaf(x0 * mag2, x0, -3.0f);
        K = e * (x0 * -0.5f);
    } else {
        K = 1.0f / sqrtf(mag2);
    }
    b[0] = K * 1.0f; b[1] = K * 0.0f; b[2] = K * 0.0f;
    a[0] = -r2; a[1] = c_wn * two_r; a[2] = 1.0f;
}

void fmo_default_controls(fmo_controls* c) {
    c->audio_out = FMO_AUDIO_STEREO;
    c->audio_stereo_mix_factor = 1.0f;
    c->use_deemphasis = 0;
    c->deemphasis_tus = 1;
    c->lpr_cutoff_hz = 15000;
    c->lmr_cutoff_hz = 15000;
}

static float design_cutoff_k(float fc, float fs) {
    /* reference broadcast_fm_demod.cpp:332-334,360-363 */
    float k = fc / (fs / 2.0f);
    return clampf(k, 0.01f, 0.99f);
}

static void design_deemphasis(float* b, float* a, int tus) {
    /* reference broadcast_fm_demod.cpp:337-352; -ffast-math folds 1/(2 pi Tus 1e-6)/(Fs/2) into (1e6/pi)/(Tus*Fs) */
    float k = MEGA_OVER_PI_F / ((float)tus * 128000.0f);
    k = clampf(k, 0.01f, 0.99f);
    fmo_design_iir_lpf(b, a, k);
}

void fmo_design(fmo_coeffs* k, int fs_baseband, const fmo_controls* c, int rsqrt_mode) {
    memset(k, 0, sizeof(*k));
    k->fs_baseband = fs_baseband;
    k->m_fm_in = fs_baseband / 256000;
    /* reference broadcast_fm_demod.cpp:129-157; rolloff factor 0.95 */
    if (k->m_fm_in > 1) {
        const float kk = (128000.0f / ((float)fs_baseband / 2.0f)) * 0.95f;
        fmo_design_fir_lpf(k->b_fm_in, 64, kk);
    }
    fmo_design_fir_lpf(k->b_fm_out, 64, (64000.0f / (256000.0f / 2.0f)) * 0.95f);
    fmo_design_hilbert(k->b_hilbert, 65);
    fmo_design_iir_peak(k->pilot_b, k->pilot_a, 19000.0f / (128000.0f / 2.0f), 0.9999f, rsqrt_mode);
    fmo_design_iir_lpf(k->pll_lpf_b, k->pll_lpf_a, 100.0f / (128000.0f / 2.0f));
    design_deemphasis(k->deemph_b, k->deemph_a, c->deemphasis_tus);
    fmo_design_fir_lpf(k->b_lpr, 128, design_cutoff_k((float)c->lpr_cutoff_hz, 128000.0f));
    fmo_design_fir_lpf(k->b_lmr, 128, design_cutoff_k((float)c->lmr_cutoff_hz, 128000.0f));
    fmo_design_fir_lpf(k->b_rds, 128, 2000.0f / (128000.0f / 2.0f));
    /* reference bpsk_synchroniser.cpp:26-48 */
    fmo_design_iir_lpf(k->ted_lpf_b, k->ted_lpf_a, 1500.0f / (16000.0f / 2.0f));
    fmo_design_iir_lpf(k->bpsk_lpf_b, k->bpsk_lpf_a, 10.0f / (16000.0f / 2.0f));
    /* reference fm_demod.cpp:35-37 as compiled: A = (1/(4 pi)) / (Fd/Fs) */
    k->fm_gain = INV_4PI_F / (75e3f / 256000.0f);
}

/* ------------------------------------------------------------------------------------------
 * PLL mixer / TED clock — reference src/fm_demod/pll_mixer.cpp:12-21, ted_clock.cpp:18-44
 * ---------------------------------------------------------------------------------------- */
typedef struct { float KTs, yn, phase_error, phase_error_gain, f_center, f_gain; } pll_mixer_t;

static inline float pll_mixer_update(pll_mixer_t* m) {
    float control = m->phase_error * m->phase_error_gain;
    control = clampf(control, -1.0f, 1.0f);
    const float freq = fmaf(control, m->f_gain, m->f_center);
    const float y = fmaf(freq, m->KTs, m->yn);
    const float t = y - round_half_away(y);
    m->yn = t;
    return t;
}

typedef struct { float KTs, yn, phase_error, phase_error_gain, fcenter, fgain; } ted_clock_t;

static inline float ted_timing_error(const ted_clock_t* c) {
    const float yn = c->yn;
    float err = yn + yn;
    if (yn > 0.5f) err = err - 2.0f;
    return err;
}

static inline int ted_update(ted_clock_t* c) {
    float control = c->phase_error * c->phase_error_gain;
    control = clampf(control, -1.0f, 1.0f);
    const float freq = fmaf(control, c->fgain, c->fcenter);
    const float d = freq * c->KTs;
    const float y = d + c->yn;
    const float thr = fmaf(-d, 0.5f, 1.0f);
    if (thr > y) { c->yn = y; return 0; }
    c->yn = 0.0f;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * BPSK synchroniser — reference src/fm_demod/bpsk_synchroniser.cpp:12-186
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int block_size;
    float zcd_xn; int cooldown_n, cooldown_remain;
    ted_clock_t ted_clock;
    float int_ted_KTs, int_ted_yn, ted_prev_phase_error, ted_kp;
    float ted_xn[2], ted_yn[2];
    float dump_KTs; fmo_cf32 dump_yn;
    pll_mixer_t mixer;
    float int_pll_KTs, int_pll_yn, pll_prev_phase_error, pll_kp;
    float pll_xn[2], pll_yn[2];
    /* traces */
    fmo_cf32 *pll_sym, *int_dump; uint8_t *zcd, *trig; float *ted_raw, *ted_pi, *pll_raw, *pll_pi;
    float *zcd_f, *trig_f;   /* the two bool traces (GetZeroCrossings, GetIntDumpTriggers :79-80) once more as 0 / 1 floats for fmo_get */
} bpsk_t;

static void bpsk_init(bpsk_t* s, int block_size) {
    memset(s, 0, sizeof(*s));
    s->block_size = block_size;
    const float Fs = 16e3f, Ts = 1.0f / Fs, Fsymbol = 2e3f;
    const int samples_per_symbol = (int)roundf(Fs / Fsymbol);
    s->cooldown_n = samples_per_symbol / 2;
    s->dump_KTs = 1.0f / (0.5f * (float)samples_per_symbol * 1.0f);
    s->ted_clock.KTs = Ts; s->ted_clock.fcenter = Fsymbol; s->ted_clock.fgain = 1.5e3f;
    s->ted_clock.phase_error_gain = 1.0f;
    s->mixer.f_center = 0.0f; s->mixer.f_gain = 10.0f; s->mixer.KTs = Ts; s->mixer.phase_error_gain = 1.0f;
    const float kk = Fsymbol / Fs;
    s->int_ted_KTs = 10.0f * Ts * kk; s->ted_kp = 0.3f;
    s->int_pll_KTs = 10.0f * Ts * kk; s->pll_kp = 0.3f;
    s->pll_sym = (fmo_cf32*)calloc(block_size, sizeof(fmo_cf32));
    s->int_dump = (fmo_cf32*)calloc(block_size, sizeof(fmo_cf32));
    s->zcd = (uint8_t*)calloc(block_size, 1); s->trig = (uint8_t*)calloc(block_size, 1);
    s->ted_raw = (float*)calloc(block_size, 4); s->ted_pi = (float*)calloc(block_size, 4);
    s->pll_raw = (float*)calloc(block_size, 4); s->pll_pi = (float*)calloc(block_size, 4);
    s->zcd_f = (float*)calloc(block_size, 4); s->trig_f = (float*)calloc(block_size, 4);
}

static void bpsk_free(bpsk_t* s) {
    free(s->pll_sym); free(s->int_dump); free(s->zcd); free(s->trig);
    free(s->ted_raw); free(s->ted_pi); free(s->pll_raw); free(s->pll_pi); free(s->zcd_f); free(s->trig_f);
}

static int bpsk_process(bpsk_t* s, const fmo_coeffs* k, const fmo_cf32* x, fmo_cf32* y) {
    int n_sym = 0;
    for (int i = 0; i < s->block_size; i++) {
        /* PI controller for the carrier PLL (:107-114) */
        const float pll_lpf = iir_f32_step(k->bpsk_lpf_b, k->bpsk_lpf_a, 2, s->pll_xn, s->pll_yn, s->pll_prev_phase_error);
        s->int_pll_yn = clampf(fmaf(s->pll_prev_phase_error, s->int_pll_KTs, s->int_pll_yn), -1.0f, 1.0f);
        const float PI_pll = fmaf(pll_lpf, s->pll_kp, s->int_pll_yn);
        s->mixer.phase_error = PI_pll;
        /* phase correction (:121-125) */
        const float dt_sin = pll_mixer_update(&s->mixer);
        float dt_cos = dt_sin + 0.25f;
        dt_cos = dt_cos - round_half_away(dt_cos);
        const float ps = fmo_chebyshev_sine(dt_sin);
        const float pc = fmo_chebyshev_sine(dt_cos);
        const float p = x[i].re, q = x[i].im;
        fmo_cf32 IQ;
        IQ.re = fmaf(pc, p, -(q * ps));
        IQ.im = fmaf(p, ps, q * pc);
        /* zero crossing detector + hold-off (:128-132; zero_crossing_detector.cpp:3-8; trigger_cooldown.cpp:4-13) */
        int is_zcd = (0.0f > (IQ.im * s->zcd_xn));
        s->zcd_xn = IQ.im;
        if (is_zcd && s->cooldown_remain == 0) { s->cooldown_remain = s->cooldown_n; is_zcd = 1; }
        else { if (s->cooldown_remain > 0) s->cooldown_remain--; is_zcd = 0; }
        if (is_zcd) s->ted_prev_phase_error = ted_timing_error(&s->ted_clock);
        /* TED PI controller (:135-143) */
        const float ted_lpf = iir_f32_step(k->ted_lpf_b, k->ted_lpf_a, 2, s->ted_xn, s->ted_yn, s->ted_prev_phase_error);
        s->int_ted_yn = clampf(fmaf(s->ted_prev_phase_error, s->int_ted_KTs, s->int_ted_yn), -1.0f, 1.0f);
        const float PI_ted = fmaf(ted_lpf, s->ted_kp, s->int_ted_yn);
        s->ted_clock.phase_error = -PI_ted;
        /* integrate and dump (:146-171) */
        s->dump_yn.re = fmaf(s->dump_KTs, IQ.re, s->dump_yn.re);
        s->dump_yn.im = fmaf(s->dump_KTs, IQ.im, s->dump_yn.im);
        const int is_ted = ted_update(&s->ted_clock);
        if (is_ted) {
            const fmo_cf32 sym = s->dump_yn;
            s->dump_yn.re = 0.0f; s->dump_yn.im = 0.0f;
            const float ph = atan2f(sym.im, sym.re);
            const float est = (ph > 0.0f) ? (HALF_PI_F - ph) : (-HALF_PI_F - ph);
            s->pll_prev_phase_error = est * TWO_OVER_PI_F;
            y[n_sym++] = sym;
        }
        s->pll_sym[i] = IQ; s->zcd[i] = (uint8_t)is_zcd; s->trig[i] = (uint8_t)is_ted;
        s->zcd_f[i] = (float)is_zcd; s->trig_f[i] = (float)is_ted;
        s->ted_raw[i] = s->ted_prev_phase_error; s->ted_pi[i] = PI_ted;
        s->pll_raw[i] = s->pll_prev_phase_error; s->pll_pi[i] = PI_pll;
        s->int_dump[i] = s->dump_yn;
    }
    return n_sym;
}

/* ------------------------------------------------------------------------------------------
 * the chain — reference src/fm_demod/broadcast_fm_demod.cpp:59-585
 * ---------------------------------------------------------------------------------------- */
struct fmo_demod {
    int block_size, n_fm_in, n_fm_out, n_rds, n_audio;
    fmo_coeffs k;
    fmo_controls ctl;
    int dirty_deemph, dirty_lpr, dirty_lmr, coeffs_overridden;
    /* state */
    fmo_cf32 h_fm_in[64];
    float prev_theta;
    float h_fm_out[64];
    float deemph_xn[2], deemph_yn[2];
    float h_hilbert[65];
    fmo_cf32 pilot_xn[3], pilot_yn[3];
    float agc_pilot_gain;
    float pll_xn[2], pll_yn[2], pll_int_yn, pll_int_KTs, pll_prev_err, pll_kp;
    pll_mixer_t pll_mixer;
    fmo_cf32 h_lpr[128], h_lmr[128], h_rds[128];
    float lmr_phase_error;
    float agc_rds_gain;
    bpsk_t bpsk;
    int n_sym;
    /* buffers */
    fmo_cf32 *in_f32, *fm_in, *fm_out_iq, *pilot, *pll, *temp_pll, *temp_audio, *rds, *rds_raw_sym;
    float *fm_demod, *fm_out, *pll_dt, *pll_raw, *pll_pi, *lpr, *lmr, *rds_sym, *audio;
};

fmo_demod* fmo_create(int block_size, int fs_baseband) {
    if (fs_baseband != 256000 && fs_baseband != 1024000 && fs_baseband != 2048000) return NULL;
    const int m = fs_baseband / 256000;
    if (block_size <= 0 || (block_size % (m * 16)) != 0) return NULL;
    fmo_demod* d = (fmo_demod*)calloc(1, sizeof(fmo_demod));
    d->block_size = block_size;
    d->n_fm_in = block_size / m;
    d->n_fm_out = d->n_fm_in / 2;
    d->n_rds = d->n_fm_out / 8;
    d->n_audio = d->n_fm_out / 4;
    fmo_default_controls(&d->ctl);
    fmo_design(&d->k, fs_baseband, &d->ctl, 1);
    d->dirty_deemph = d->dirty_lpr = d->dirty_lmr = 1; /* SetValue() in the ctor (:189,:248,:260) */
    d->agc_pilot_gain = 0.1f;
    d->agc_rds_gain = 0.1f;
    const float Ts = 1.0f / 128000.0f;
    d->pll_mixer.f_center = -19000.0f; d->pll_mixer.f_gain = -100.0f; d->pll_mixer.KTs = Ts; d->pll_mixer.phase_error_gain = 1.0f;
    d->pll_int_KTs = 0.1f * Ts; d->pll_kp = 0.01f;
    bpsk_init(&d->bpsk, d->n_rds);
    d->in_f32 = (fmo_cf32*)calloc(block_size, sizeof(fmo_cf32));
    d->fm_in = (fmo_cf32*)calloc(d->n_fm_in, sizeof(fmo_cf32));
    d->fm_demod = (float*)calloc(d->n_fm_in, 4);
    d->fm_out = (float*)calloc(d->n_fm_out, 4);
    d->fm_out_iq = (fmo_cf32*)calloc(d->n_fm_out, sizeof(fmo_cf32));
    d->pilot = (fmo_cf32*)calloc(d->n_fm_out, sizeof(fmo_cf32));
    d->pll = (fmo_cf32*)calloc(d->n_fm_out, sizeof(fmo_cf32));
    d->pll_dt = (float*)calloc(d->n_fm_out, 4);
    d->pll_raw = (float*)calloc(d->n_fm_out, 4);
    d->pll_pi = (float*)calloc(d->n_fm_out, 4);
    d->temp_pll = (fmo_cf32*)calloc(d->n_fm_out, sizeof(fmo_cf32));
    d->temp_audio = (fmo_cf32*)calloc(d->n_audio, sizeof(fmo_cf32));
    d->lpr = (float*)calloc(d->n_audio, 4);
    d->lmr = (float*)calloc(d->n_audio, 4);
    d->rds = (fmo_cf32*)calloc(d->n_rds, sizeof(fmo_cf32));
    d->rds_raw_sym = (fmo_cf32*)calloc(d->n_rds, sizeof(fmo_cf32));
    d->rds_sym = (float*)calloc(d->n_rds, 4);
    d->audio = (float*)calloc((size_t)d->n_audio * 2, 4);
    return d;
}

void fmo_destroy(fmo_demod* d) {
    if (!d) return;
    bpsk_free(&d->bpsk);
    free(d->in_f32); free(d->fm_in); free(d->fm_demod); free(d->fm_out); free(d->fm_out_iq); free(d->pilot);
    free(d->pll); free(d->pll_dt); free(d->pll_raw); free(d->pll_pi); free(d->temp_pll); free(d->temp_audio);
    free(d->lpr); free(d->lmr); free(d->rds); free(d->rds_raw_sym); free(d->rds_sym); free(d->audio);
    free(d);
}

void fmo_set_controls(fmo_demod* d, const fmo_controls* c) {
    if (c->deemphasis_tus != d->ctl.deemphasis_tus) d->dirty_deemph = 1;
    if (c->lpr_cutoff_hz != d->ctl.lpr_cutoff_hz) d->dirty_lpr = 1;
    if (c->lmr_cutoff_hz != d->ctl.lmr_cutoff_hz) d->dirty_lmr = 1;
    d->ctl = *c;
}

void fmo_set_coeffs(fmo_demod* d, const fmo_coeffs* k) { d->k = *k; d->coeffs_overridden = 1; d->dirty_deemph = d->dirty_lpr = d->dirty_lmr = 0; }
void fmo_get_coeffs(fmo_demod* d, fmo_coeffs* k) { *k = d->k; }

/* reference broadcast_fm_demod.cpp:330-389 */
static void update_filters(fmo_demod* d) {
    if (d->coeffs_overridden) return;
    if (d->dirty_deemph) { d->dirty_deemph = 0; design_deemphasis(d->k.deemph_b, d->k.deemph_a, d->ctl.deemphasis_tus); }
    if (d->dirty_lpr) { d->dirty_lpr = 0; fmo_design_fir_lpf(d->k.b_lpr, 128, design_cutoff_k((float)d->ctl.lpr_cutoff_hz, 128000.0f)); }
    if (d->dirty_lmr) { d->dirty_lmr = 0; fmo_design_fir_lpf(d->k.b_lmr, 128, design_cutoff_k((float)d->ctl.lmr_cutoff_hz, 128000.0f)); }
}

/* :391-416 */
static void run_fm_demodulate(fmo_demod* d, const fmo_cf32* x) {
    const fmo_coeffs* k = &d->k;
    if (k->m_fm_in > 1) fmo_decim_c32(d->h_fm_in, k->b_fm_in, 64, k->m_fm_in, x, d->fm_in, d->n_fm_in);
    else memcpy(d->fm_in, x, sizeof(fmo_cf32) * (size_t)d->n_fm_in);
    fmo_discriminator(&d->prev_theta, k->fm_gain, d->fm_in, d->fm_demod, d->n_fm_in);
    fmo_decim_f32(d->h_fm_out, k->b_fm_out, 64, 2, d->fm_demod, d->fm_out, d->n_fm_out);
    if (d->ctl.use_deemphasis) fmo_iir_f32(k->deemph_b, k->deemph_a, 2, d->deemph_xn, d->deemph_yn, d->fm_out, d->fm_out, d->n_fm_out);
    fmo_hilbert(d->h_hilbert, k->b_hilbert, d->fm_out, d->fm_out_iq, d->n_fm_out);
}

/* :418-461 */
static void lock_onto_pilot(fmo_demod* d) {
    const fmo_coeffs* k = &d->k;
    const int N = d->n_fm_out;
    fmo_iir_c32(k->pilot_b, k->pilot_a, 3, d->pilot_xn, d->pilot_yn, d->fm_out_iq, d->pilot, N);
    fmo_agc(&d->agc_pilot_gain, 1.0f, 0.2f, d->pilot, d->pilot, N);
    for (int i = 0; i < N; i++) {
        const float lpf = iir_f32_step(k->pll_lpf_b, k->pll_lpf_a, 2, d->pll_xn, d->pll_yn, d->pll_prev_err);
        const float P = lpf * d->pll_kp;
        d->pll_int_yn = clampf(fmaf(d->pll_prev_err, d->pll_int_KTs, d->pll_int_yn), -1.0f, 1.0f);
        const float PI_error = d->pll_int_yn + P;
        d->pll_mixer.phase_error = PI_error;
        const float dt_sin = pll_mixer_update(&d->pll_mixer);
        float dt_cos = dt_sin + 0.25f;
        dt_cos = dt_cos - round_half_away(dt_cos);
        const float ps = fmo_chebyshev_sine(dt_sin);
        const float pc = fmo_chebyshev_sine(dt_cos);
        const float p = d->pilot[i].re, q = d->pilot[i].im;
        const float res_im = fmaf(ps, p, q * pc);
        const float res_re = fmaf(p, pc, -(q * ps));
        d->pll_prev_err = atan2f(res_im, res_re);
        d->pll_dt[i] = dt_sin;
        d->pll[i].re = pc; d->pll[i].im = ps;
        d->pll_raw[i] = d->pll_prev_err;
        d->pll_pi[i] = PI_error;
    }
}

/* :463-536 */
static void extract_components(fmo_demod* d) {
    const fmo_coeffs* k = &d->k;
    const int N = d->n_fm_out, NA = d->n_audio;
    fmo_decim_c32(d->h_lpr, k->b_lpr, 128, 4, d->fm_out_iq, d->temp_audio, NA);
    for (int i = 0; i < NA; i++) d->lpr[i] = d->temp_audio[i].re;
    fmo_harmonic_mix(d->pll_dt, d->fm_out_iq, d->temp_pll, N, 38000.0f / 19000.0f, d->lmr_phase_error);
    fmo_decim_c32(d->h_lmr, k->b_lmr, 128, 4, d->temp_pll, d->temp_audio, NA);
    {
        float sum = 0.0f;
        int total = 0;
        for (int i = 0; i < NA; i += 10) {
            const float ph = atan2f(d->temp_audio[i].im, d->temp_audio[i].re);
            const float est = (ph > 0.0f) ? (HALF_PI_F - ph) : (-HALF_PI_F - ph);
            sum = sum + est;
            total++;
        }
        const float avg = sum / (float)total;
        const float acc = fmaf(avg, 0.1f, d->lmr_phase_error);
        d->lmr_phase_error = fmodf(acc, TWO_PI_F);
    }
    for (int i = 0; i < NA; i++) d->lmr[i] = d->temp_audio[i].im;
    fmo_harmonic_mix(d->pll_dt, d->fm_out_iq, d->temp_pll, N, 57000.0f / 19000.0f, 0.0f);
    fmo_decim_c32(d->h_rds, k->b_rds, 128, 8, d->temp_pll, d->rds, d->n_rds);
}

/* :538-547 */
static void synchronise_rds(fmo_demod* d) {
    fmo_agc(&d->agc_rds_gain, 0.5f, 0.2f, d->rds, d->rds, d->n_rds);
    d->n_sym = bpsk_process(&d->bpsk, &d->k, d->rds, d->rds_raw_sym);
    for (int i = 0; i < d->n_sym; i++) d->rds_sym[i] = d->rds_raw_sym[i].im;
}

/* :549-585 */
static void mix_audio(fmo_demod* d) {
    const int N = d->n_audio;
    const float kmix = d->ctl.audio_stereo_mix_factor;
    for (int i = 0; i < N; i++) {
        float l, r;
        if (d->ctl.audio_out == FMO_AUDIO_STEREO) {
            l = fmaf(d->lmr[i], kmix, d->lpr[i]);
            r = fmaf(-d->lmr[i], kmix, d->lpr[i]);
        } else if (d->ctl.audio_out == FMO_AUDIO_LMR) {
            l = r = d->lmr[i];
        } else {
            l = r = d->lpr[i];
        }
        d->audio[2 * i] = l + l;
        d->audio[2 * i + 1] = r + r;
    }
}

int fmo_process_cf32(fmo_demod* d, const float* iq, int n) {
    if (n != d->block_size) return -1;  /* reference :311-313: silently drops the block */
    update_filters(d);
    run_fm_demodulate(d, (const fmo_cf32*)iq);
    lock_onto_pilot(d);
    extract_components(d);
    synchronise_rds(d);
    mix_audio(d);
    return 0;
}

/* reference src/app.cpp:56-65 */
int fmo_process_u8(fmo_demod* d, const uint8_t* iq, int n) {
    if (n != d->block_size) return -1;
    for (int i = 0; i < n; i++) {
        d->in_f32[i].re = (float)iq[2 * i] - 127.0f;
        d->in_f32[i].im = (float)iq[2 * i + 1] - 127.0f;
    }
    return fmo_process_cf32(d, (const float*)d->in_f32, n);
}

int fmo_rds_symbol_count(fmo_demod* d) { return d->n_sym; }

const float* fmo_get(fmo_demod* d, const char* name, int* n) {
#define RET(p, cnt) do { *n = (cnt); return (const float*)(p); } while (0)
    if (!strcmp(name, "fm_in")) RET(d->fm_in, 2 * d->n_fm_in);
    if (!strcmp(name, "fm_demod")) RET(d->fm_demod, d->n_fm_in);
    if (!strcmp(name, "fm_out")) RET(d->fm_out, d->n_fm_out);
    if (!strcmp(name, "fm_out_iq")) RET(d->fm_out_iq, 2 * d->n_fm_out);
    if (!strcmp(name, "pilot")) RET(d->pilot, 2 * d->n_fm_out);
    if (!strcmp(name, "pll_dt")) RET(d->pll_dt, d->n_fm_out);
    if (!strcmp(name, "pll")) RET(d->pll, 2 * d->n_fm_out);
    if (!strcmp(name, "pll_raw_err")) RET(d->pll_raw, d->n_fm_out);
    if (!strcmp(name, "pll_pi_err")) RET(d->pll_pi, d->n_fm_out);
    if (!strcmp(name, "lpr")) RET(d->lpr, d->n_audio);
    if (!strcmp(name, "lmr")) RET(d->lmr, d->n_audio);
    if (!strcmp(name, "rds")) RET(d->rds, 2 * d->n_rds);
    if (!strcmp(name, "rds_raw_sym")) RET(d->rds_raw_sym, 2 * d->n_sym);
    if (!strcmp(name, "rds_sym")) RET(d->rds_sym, d->n_sym);
    if (!strcmp(name, "audio")) RET(d->audio, 2 * d->n_audio);
    if (!strcmp(name, "lmr_phase")) RET(&d->lmr_phase_error, 1);
    if (!strcmp(name, "agc_pilot_gain")) RET(&d->agc_pilot_gain, 1);
    if (!strcmp(name, "agc_rds_gain")) RET(&d->agc_rds_gain, 1);
    if (!strcmp(name, "bpsk_pll_sym")) RET(d->bpsk.pll_sym, 2 * d->n_rds);
    if (!strcmp(name, "bpsk_intdump")) RET(d->bpsk.int_dump, 2 * d->n_rds);
    if (!strcmp(name, "bpsk_ted_raw")) RET(d->bpsk.ted_raw, d->n_rds);
    if (!strcmp(name, "bpsk_ted_pi")) RET(d->bpsk.ted_pi, d->n_rds);
    if (!strcmp(name, "bpsk_pll_raw")) RET(d->bpsk.pll_raw, d->n_rds);
    if (!strcmp(name, "bpsk_pll_pi")) RET(d->bpsk.pll_pi, d->n_rds);
    if (!strcmp(name, "bpsk_zcd")) RET(d->bpsk.zcd_f, d->n_rds);
    if (!strcmp(name, "bpsk_trig")) RET(d->bpsk.trig_f, d->n_rds);
#undef RET
    *n = 0;
    return NULL;
}

/* ------------------------------------------------------------------------------------------
 * Differential Manchester decoder — reference src/rds_decoder/differential_manchester_decoder.h:25-60
 * ---------------------------------------------------------------------------------------- */
void fmo_manchester_init(fmo_manchester* m) { memset(m, 0, sizeof(*m)); }

int fmo_manchester_push(fmo_manchester* m, const float* sym, int n, uint8_t* out, int out_cap) {
    int written = 0;
    for (int i = 0; i < n; i++) {
        m->is_read_bit = !m->is_read_bit;
        if (!m->is_read_bit) continue;
        const int curr = (sym[i] > 0.0f);
        const int bit = curr ^ m->prev_bit;
        m->prev_bit = curr;
        if (m->bit_index == 0) m->buf[m->byte_index] = 0;
        m->buf[m->byte_index] |= (uint8_t)((bit & 1) << (7 - m->bit_index));
        m->bit_index++;
        m->byte_index += m->bit_index / 8;
        m->bit_index = m->bit_index % 8;
        if (m->byte_index == 16) {
            m->byte_index = 0;
            if (written + 16 <= out_cap) { memcpy(out + written, m->buf, 16); written += 16; }
        }
    }
    return written;
}
