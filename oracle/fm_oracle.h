/* TEST INFRASTRUCTURE — CPU restatement of the reference FM demodulation path.
 *
 * This is the parity oracle: a plain-C, single-threaded, scalar restatement of
 * williamyang98/FM-Radio's Broadcast_FM_Demod::Process and everything it calls
 * (reference src/fm_demod/broadcast_fm_demod.cpp:309-328).  It is NOT product code: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker.  The product path is the HIP library behind include/fmdemod.h.
 *
 * Pinning: every operation order in fm_oracle.c (which products are rounded, which are
 * fused, how sums are associated) follows the reference's own `gcc` preset build
 * (-O2 -ffast-math, AVX2+FMA) as read from its disassembly, and tests/test_oracle_vs_ref.py
 * checks the restatement BIT-FOR-BIT against that build (oracle/_ref/fm_ref_dump) on
 * synthetic captures; tests/golden/ holds vectors dumped from the same build.
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } fmo_cf32;

enum { FMO_AUDIO_LPR = 0, FMO_AUDIO_LMR = 1, FMO_AUDIO_STEREO = 2 };

/* reference Broadcast_FM_Demod_Controls (broadcast_fm_demod.h:64-89) */
typedef struct {
    int   audio_out;               /* FMO_AUDIO_* ; default STEREO */
    float audio_stereo_mix_factor; /* default 1.0 */
    int   use_deemphasis;          /* default 0 */
    int   deemphasis_tus;          /* default 1 (us) */
    int   lpr_cutoff_hz;           /* default 15000 */
    int   lmr_cutoff_hz;           /* default 15000 */
} fmo_controls;

/* Every designed coefficient + derived constant the chain uses. */
typedef struct {
    int   fs_baseband;       /* 256000 (no stage 1), 1024000 (M=4) or 2048000 (M=8) */
    int   m_fm_in;           /* stage-1 decimation: 1, 4 or 8 */
    float b_fm_in[64];       /* broadcast_fm_demod.cpp:133-144 */
    float b_fm_out[64];      /* :146-157 */
    float b_hilbert[65];     /* hilbert_fir_filter.h:17-21 */
    float pilot_b[3], pilot_a[3];     /* :200-213 */
    float pll_lpf_b[2], pll_lpf_a[2]; /* :215-224 */
    float deemph_b[2], deemph_a[2];   /* :337-352 */
    float b_lpr[128], b_lmr[128];     /* :354-388 */
    float b_rds[128];                 /* :263-274 */
    float ted_lpf_b[2], ted_lpf_a[2]; /* bpsk_synchroniser.cpp:26-36 */
    float bpsk_lpf_b[2], bpsk_lpf_a[2]; /* :38-48 */
    float fm_gain;           /* fm_demod.cpp:35-37 */
} fmo_coeffs;

typedef struct fmo_demod fmo_demod;

void fmo_default_controls(fmo_controls* c);
/* rsqrt_mode: 1 = reproduce the reference build's rsqrtss+Newton step for the pilot peak gain
 * (bit-exact with oracle/_ref on the same CPU), 0 = IEEE sqrt/divide. */
void fmo_design(fmo_coeffs* k, int fs_baseband, const fmo_controls* c, int rsqrt_mode);

fmo_demod* fmo_create(int block_size, int fs_baseband);
void fmo_destroy(fmo_demod* d);
void fmo_set_controls(fmo_demod* d, const fmo_controls* c);    /* marks filters dirty like SetValue() */
void fmo_set_coeffs(fmo_demod* d, const fmo_coeffs* k);        /* override designed coefficients */
void fmo_get_coeffs(fmo_demod* d, fmo_coeffs* k);
int  fmo_process_cf32(fmo_demod* d, const float* iq, int n);   /* 0 ok, -1 wrong size (no output, like the reference) */
int  fmo_process_u8(fmo_demod* d, const uint8_t* iq, int n);   /* App::Run conversion then Process */

/* views valid until the next process call; *n receives the element count */
const float* fmo_get(fmo_demod* d, const char* name, int* n);
int fmo_rds_symbol_count(fmo_demod* d);

/* Differential Manchester decoder (reference rds_decoder/differential_manchester_decoder.h) */
typedef struct {
    uint8_t buf[16];
    int byte_index, bit_index, is_read_bit, prev_bit;
} fmo_manchester;
void fmo_manchester_init(fmo_manchester* m);
/* returns number of bytes appended to out (multiples of 16) */
int fmo_manchester_push(fmo_manchester* m, const float* sym, int n, uint8_t* out, int out_cap);

/* primitives exported for unit tests */
float fmo_chebyshev_sine(float x);
void fmo_atan2f_array(const float* y, const float* x, float* out, long n);
float fmo_dot_f32(const float* x, const float* b, int n);
fmo_cf32 fmo_dot_c32(const fmo_cf32* x, const float* b, int n);
void fmo_decim_c32(fmo_cf32* hist, const float* b, int nn, int m, const fmo_cf32* x, fmo_cf32* y, int n_out);
void fmo_decim_f32(float* hist, const float* b, int nn, int m, const float* x, float* y, int n_out);
void fmo_hilbert(float* hist65, const float* b65, const float* x, fmo_cf32* y, int n);
void fmo_iir_c32(const float* b, const float* a, int k, fmo_cf32* xn, fmo_cf32* yn, const fmo_cf32* x, fmo_cf32* y, int n);
void fmo_iir_f32(const float* b, const float* a, int k, float* xn, float* yn, const float* x, float* y, int n);
float fmo_agc(float* gain, float target_power, float beta, const fmo_cf32* x, fmo_cf32* y, int n);
void fmo_discriminator(float* prev_theta, float gain, const fmo_cf32* x, float* y, int n);
void fmo_harmonic_mix(const float* dt, const fmo_cf32* x, fmo_cf32* y, int n, float harmonic, float offset);
void fmo_design_fir_lpf(float* b, int n, float k);
void fmo_design_hilbert(float* b, int n);
void fmo_design_iir_lpf(float* b, float* a, float k);
void fmo_design_iir_peak(float* b, float* a, float k, float r, int rsqrt_mode);

#ifdef __cplusplus
}
#endif
#endif
