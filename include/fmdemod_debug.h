/* fmdemod_debug.h — test, self-check and profiling hooks of libfmdemod.so.
 *
 * NOT part of the drop-in boundary (include/fmdemod.h): nothing here has a counterpart in the reference's demodulator API.
 * tests/ use the self-tests to compare the device math with the host libm the reference links; bench.py uses the profiling
 * hooks for its roofline figures.  Results of the demodulator never depend on any of these calls.
 */
#ifndef FMDEMOD_DEBUG_H
#define FMDEMOD_DEBUG_H

#include "fmdemod.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-kernel timing with HIP events recorded on the processing stream (for bench.py's roofline figures).
 * on = 1: every fmd_process_* call brackets each kernel of the sequence with events; on = 2: k_pilot_pll (the dominant
 * kernel) of every block, the other kernels of every 4th block — the brackets are extra queue packets between dependent
 * kernels and cost the pipelined step a few percent; fmd_profile_read synchronises, accumulates and clears them. */
typedef struct {
    char   name[32];     /* kernel name as it appears in rocprofv3 kernel traces (prefix match) */
    double total_ms;     /* sum of launch durations since the last read */
    int    launches;
} fmd_kernel_time;
/* Self-test hook: evaluates the kernels' atan2f on the device for n host-side (y, x) pairs, so tests can compare
 * the device math bit-for-bit with the host libm the reference links (std::atan2, reference fm_demod.cpp:40). */
int fmd_selftest_atan2(const float* y, const float* x, float* out, size_t n);
/* Same for the table-driven form the discriminator (k_front) uses: identical values, fewer issued instructions. */
int fmd_selftest_atan2_table(const float* y, const float* x, float* out, size_t n);
/* ... and its variant for u8 IQ at 256 kSa/s, whose operands are the integers -127..128 (only 0/0 is special there). */
int fmd_selftest_atan2_table_u8(const float* y, const float* x, float* out, size_t n);
/* Same for the short form k_pilot_pll's phase detector uses on a locked loop: out[i] is only meaningful where ok[i] != 0, and
 * there it must equal atan2f(y[i], x[i]) bit-for-bit (DESIGN.md "Pilot PLL"). */
int fmd_selftest_atan2_small(const float* y, const float* x, float* out, uint8_t* ok, size_t n);

/* The FMD_FLAG_FAST_MATH primitives evaluated on the device, for accuracy checks against float64 on the host:
 * kind 0: out = fast atan2(a, b); kind 1: out = sin(2 pi a) (hardware, argument in turns); kind 2: out = cos(2 pi a);
 * kind 3: out = atan2(a, b) / 2 pi, the six-coefficient form of the discriminator and the pilot loop's phase detector. */
int fmd_selftest_fast_math(int kind, const float* a, const float* b, float* out, size_t n);

/* Host-only (no GPU needed): the tables of the tolerance mode's span-wise pilot PLL (fm-radio_amd/csrc/fmd_kernels.h PllSpanTab) as
 * the library designs them for fs_baseband — w [5][128] weights of the held-frequency error sequence (rows: loop filter, integrator,
 * phase deviation at samples 41, 84, 127), s [5][8] weights of the state (lpf, I, e1, e2, r0), minv [3][4] cubic fit, misc[0] = the
 * quadrature factor, misc[1] = kappa.  tests/test_span_design.py checks them against an independent float64 restatement. */
int fmd_design_pll_span(int fs_baseband, float* w, float* s, float* minv, float* misc2);
/* ... and of the round-4 form that evaluates the loop's phase detector at 8 points of a span (PllSparseTab, k_pll_sparse): taps [2][32]
 * (real, imaginary weight of a point's 32 input samples), cplx [19][2] = rot[8], scan[3], carry[8] as (re, im), rows [2][8] = wsum, wmom
 * (5 used each), sw [5][132] suffix sums of the weight rows, misc8 = phi0, inv_s2, nbar, kappa, pw_scale, kap2 (re, im), 0.
 * tests/test_span_design.py checks them against a float64 model of the reference's peak filter and loop. */
int fmd_design_pll_sparse(int fs_baseband, float* taps, float* cplx, float* rows, float* sw, float* misc8);

/* Taps of k_extract_bp's composite band-pass FIRs (fmd_kernels_bp.inc: the harmonic mixer's carrier and the Hilbert FIR folded into the
 * decimating FIR): g2 [2][192] = (real, imaginary) taps of the L-R FIR for an L-R cut-off of cutoff_hz, g3 [2][192] of the RDS FIR.
 * tests/test_bandpass_design.py checks them against the reference's order of operations in float64. */
int fmd_design_extract_bp(int fs_baseband, int cutoff_hz, float* g2, float* g3);

/* The tolerance mode's table for the discriminator's wrap at exactly half a turn on u8 captures (fmd_kernels.hip wrap_tie_u8): 65536 bits,
 * bit (y_raw << 8 | x_raw) = 1 when the reference's wrapped phase difference from sample (x_raw, y_raw) to one in exactly the opposite direction
 * is +pi.  tests/test_wrap_tie.py checks it against glibc's atan2f. */
int fmd_design_wrap_tie(uint32_t* bits2048);

/* Counters of k_pilot_pll's frequency speculation since creation / the last reset (DESIGN.md "Pilot PLL"):
 * out8[0] = 128-sample chunks, summed over wavefronts (4 channels each); out8[1] = of those, chunks run with the plain serial
 * iteration (wavefront out of lock); out8[2] = spans redone with the reference forms (a short form outside its domain);
 * out8[3] = spans, out8[4] = samples committed, both summed over channels (ratio = samples per 16-sample span);
 * out8[5] = spans run in the sequence form (a loop out of lock: the span speculates on the sequence of frequency words), summed over wavefronts;
 * out8[6], out8[7] = shader-clock cycles and 100 MHz real-time ticks of one wavefront per launch (ratio x 100 = core MHz).
 * Results never depend on any of them; they explain k_pilot_pll's duration. */
int fmd_get_spec_stats(fmd_handle h, uint64_t* out8, int reset);

/* on = 0: off; 1: timing events on every kernel of every block; 2: the dominant kernel every block, the others every 4th;
 * 3: every kernel of every 4th block plus the dominant kernel of the block behind it (what bench.py uses: ~1 % of the step) */
int fmd_profile_enable(fmd_handle h, int on);
/* 1.024 / 2.048 MSa/s, tolerance mode: on = 1 runs the first decimator and the front end as the two kernels they were (k_predecim_mfma,
 * k_front_mfma with fm_in through HBM) instead of k_front_pre_mfma; results are bit-identical either way (tests/test_gpu_fast.py).
 * Between blocks only. */
int fmd_debug_split_front(fmd_handle h, int on);
int fmd_profile_read(fmd_handle h, fmd_kernel_time* out, int cap, int* n_out);
/* Tolerance mode, 256 kSa/s cf32: on = 1 runs a steady block's front end, pilot stage and extract stage as ONE launch (k_chain,
 * fm-radio_amd/csrc/fmd_kernels_chain.inc: fm_out, the pilot points' sums and the span cubics stay in LDS) instead of the three launches
 * that are the default.  Round 6's measured A/B (profiles/round6/chain_*): 22 % fewer HBM bytes (832 MB a block
 * against 1069), correct — within 1.1e-6 RMS of the three-launch form's audio, the same distance from the oracle — and SLOWER (0.268 ms against
 * 0.237 without the RDS stage, 0.44 against 0.25 with it): two 80 KB workgroups per CU are all the LDS admits, ten wavefronts a CU do not
 * hide the chain's latencies, and the RDS stage's workgroups take slots from a grid that needs every one of them.  Kept as a development switch and as the parity test of both forms.
 * Between blocks only; both forms leave the same histories, so a handle may change between them at any block.
 * fmd_debug_chain_blocks: how many blocks since create / reset ran as k_chain. */
int fmd_debug_set_chain(fmd_handle h, int on);
/* Exact mode, test hook: the batch sizes up to which the pilot PLL runs as the time-parallel kernel with 16 lanes a station (default 3584) and as the
 * time-parallel kernel at all (default 7168; the low-work kernel above).  Beyond either the choice follows what is out of lock:
 * 8 lanes / the low-work kernel while every loop holds lock, 16 lanes (up to 4096 stations) / the time-parallel kernel while some do not.  With small
 * values a small batch exercises the switches (tests/test_gpu_parity.py); results are bit-identical whatever kernel runs. */
int fmd_debug_pll_adaptive(fmd_handle h, int k16_max_channels, int time_parallel_max_channels);
/* Tolerance mode: k_extract_bp with two stations per workgroup (the tap tables and the block edge's matrix fetched once for both; results bit-identical):
 * 0 = where it pays (3072 stations and more with equal cut-offs, the default), 1 = wherever possible (so that tests reach it with a few stations), 2 = never. */
int fmd_debug_extract_pairing(fmd_handle h, int mode);
int fmd_debug_chain_blocks(fmd_handle h, long* blocks);

#ifdef __cplusplus
}
#endif
#endif
