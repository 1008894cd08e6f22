// Hand-written CDNA4 (gfx950) kernels of the batched broadcast-FM demodulator.
//
// One `Broadcast_FM_Demod::Process` call of the reference (src/fm_demod/broadcast_fm_demod.cpp:309-328)
// for C independent stations becomes this kernel sequence:
//
//   k_front        [parallel]  a0 u8->f32, a1 decimating FIR, a2 arctan discriminator, a3 decimating FIR,
//                              a5 Hilbert FIR  -> fm_out_iq            (LDS-staged, halo recomputed per tile)
//   k_pilot_power  [serial]    a6 pilot peak IIR + a7 AGC power sum     -> pilot   (lane per channel, LDS transpose)
//   k_pilot_pll    [serial*]   a7 gain, a8 PLL loop                     -> pll_dt  (*time-parallel under frequency speculation:
//                              16 or 8 lanes = consecutive samples of one channel; k_pilot_pll_pairs = low-work variant)
//   k_extract      [parallel]  a9 x2/x3 harmonic mixers fused into a10/a12 decimating FIRs, a11 phase
//                              estimates, a15 audio mix                -> audio, rds, lmr_est
//   k_lmr_phase    [tiny]      a11 phase integrate (the next block's k_extract needs it)
//   k_rds_sync     [serial]    a13 AGC, a14 BPSK synchroniser, Manchester decode   (own stream)
//
// Tolerance mode (FMD_FLAG_FAST_MATH) — the same path on four kernels:
//
//   k_front_mfma   [parallel]  a0-a3 (arctangent in turns, decimating FIR on the matrix cores), optional a4 inside the tile, the pilot
//                              stage's column sums                                             -> fm_out plane, pv_pl
//   k_front_pre_mfma           1.024 / 2.048 MSa/s: a0, a1 (first decimator on the matrix cores, phases kept in LDS), then the same
//                              (k_predecim_mfma + k_front_mfma<float> with fm_in through HBM when a station is de-emphasised)
//   k_pll_sparse   [serial]    a6 peak filter as a decimated one-pole, a7 AGC state, a8 the loop at eight points per 128-sample span
//                              (fmd_kernels_sparse.inc; the first workgroups of the next block's front-end launch on the deferred
//                              schedule; k_pll_span of fmd_kernels_fast.inc for a station's first 8192 samples)  -> one cubic per span
//   k_extract_bp   [parallel]  a5 Hilbert FIR and a9 mixers folded into the a10 / a12 decimating FIRs (matrix cores), a11, a15, the RDS AGC's
//                              block power as per-tile partial sums                            -> audio, rds, lmr_est, rds_pow
//   k_rds_sync3    [serial]    a13, a14, Manchester decode: the loop pipelined over five wavefronts (fmd_kernels_fast.inc)
//
// Arithmetic contract of the EXACT mode: this file is compiled with -ffp-contract=off; every fused multiply-add is an
// explicit fmaf() and every sum is associated exactly as the reference's AVX2+FMA build associates it
// (8 / 4 lane accumulators per dot product, horizontal sums in the x86 order), so results are
// bit-identical to the CPU path; no MFMA there: the FIRs are VALU work staged through LDS.
#include "fmd_kernels.h"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>
#include "fmd_math.h"

// launch with the stage's timing events attached to the dispatch packet when the caller asked for them
#define FMD_LAUNCH(r, first, last, kern, grid, block, lds, s, ...)                                                   \
    do {                                                                                                             \
        hipEvent_t e0_ = (first) ? (r).t0 : nullptr, e1_ = (last) ? ((r).t1 ? (r).t1 : (r).done) : nullptr;         \
        if (e0_ || e1_) hipExtLaunchKernelGGL(kern, grid, block, lds, s, e0_, e1_, 0, __VA_ARGS__);                  \
        else hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__);                                             \
    } while (0)


namespace fmd {

static constexpr int kWave = 64;

__device__ __forceinline__ float& st(float* state, int field, int C, int c) { return state[(size_t)field * C + c]; }

#include "fmd_kernels_sparse.inc"

// =============================================================================================
// k_front — reference Run_FM_Demodulate (broadcast_fm_demod.cpp:391-416) without the optional IIR, from the 256 kSa/s stream
// fm_in on (the capture itself at 256 kSa/s, k_predecim's output at 1.024 / 2.048 MSa/s):
//   FM_Demod::Process (fm_demod.cpp:30-45)
//   PolyphaseDownsampler<float> 2 x 64 taps (f32_cum_mul.cpp:52-78)
//   Hilbert_FIR_Filter<float> 65 taps (hilbert_fir_filter.h:26-46), 33 zero taps skipped
// One workgroup = one channel x T output samples (128 kHz).  The halo each stage needs is recomputed from `tail ++ block`
// so tiles are independent.
// =============================================================================================
// TT: fm_out samples per workgroup.  A larger tile recomputes less halo (the 191-sample halo of the three cascaded stages costs
// 19 % extra discriminator work at 512, 9 % at 1024); 1024 is used whenever the block length allows it.
// (Tolerance mode, k_front_mfma: with a de-emphasised channel the IIR runs inside the tile from a zero state WU = kDeemphWarmup fm_out
// samples before the first sample the Hilbert FIR needs; its pole is at most 0.9 (75 us), so what the zero state leaves at the
// first used sample is below 0.9^128 = 1.4e-6 of the signal.  Costs WU more outputs of the two front stages per tile: 12.5 % at 1024.)
static constexpr int kDeemphWarmup = 128;
template <int TT = 512>
struct FrontGeom {
    static constexpr int T = TT;
    static constexpr int NF = T + 64;                                    // fm_out samples a tile makes (64: history of the Hilbert FIR)
    static constexpr int NW = 2 * NF + 63;                               // fm_in samples (incl. one for prev_theta)
    static constexpr int TAIL = 191;                                     // history samples of the input stream
    // LDS carve-up in floats; every sub-array starts on a 16-byte boundary (a ds_read_b64 that is only 4-byte
    // aligned is replayed at ~64 cycles per wave instruction)
    static constexpr int NWP = (NW + 3) & ~3;
    static constexpr int OFF_THETA = 0;
    static constexpr int OFF_DEM = OFF_THETA;                            // the discriminator output replaces the phases in place: 9 KB less
                                                                         // LDS per workgroup (13.7 KB), more workgroups beside the other stages' kernels
    static constexpr int OFF_FO = OFF_DEM + NWP;
    static constexpr int OFF_ATAN = (OFF_FO + NF + 7) & ~7;              // AtanTable (32-byte aligned rows)
    static constexpr int LDS_FLOATS = OFF_ATAN + kAtanTableWords;
    static_assert(OFF_DEM % 4 == 0 && OFF_FO % 4 == 0, "LDS sub-arrays must be 16-byte aligned");
};

// FMD_FLAG_FAST_MATH: k_front's two FIRs on the matrix cores.  Both have taps common to all stations, so each is a constant
// banded-Toeplitz matrix times a matrix whose columns are overlapping windows of the signal:
//   decimate-by-2, 64 taps:  Y[m][c] = y[16 c + m] = sum_t A[m][t] dem[32 c + t],  A[m][t] = h[t - 2 m]       (t < 94)
//   Hilbert, 65 taps:        Y[m][c] = im[16 c + m] = sum_t A[m][t] fo[16 c + t],   A[m][t] = b[t - m]         (t < 80)
// i.e. three v_mfma_f32_16x16x32_bf16 K-steps per 16 x 16 outputs.  fp32 operands are split into two bf16 halves (taps: rounded,
// on the host; samples: truncated, x = hi + lo + O(2^-16 x)) and a product is hi hi + lo hi + hi lo, accumulated in fp32:
// 9 MFMAs per 256 outputs and wavefront instead of 256 x 64 / 64 = 256 (128) VALU FMAs, on a pipe of their own — the VALU issue
// slots are what bounds the step.  The operand images of A (lane l: row l % 16, k = 8 (l / 16) .. + 7, layout checked
// by tools/mfma_bf16_probe.hip) come ready-made from the host (FrontMfmaTab).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
static constexpr f32x4 kZero4 = {0.f, 0.f, 0.f, 0.f};
template <int TT, int WU>
struct FrontGeomM {
    static constexpr int T = TT, NF = T + WU, NW = 2 * NF + 63, TAIL = 63 + 2 * WU;   // (no Hilbert history: k_extract_bp takes that FIR)
    static constexpr int NWB = (NW + 8 + 7) & ~7;        // bf16 elements per half (hi / lo) of the discriminator output, zero padded
    static constexpr int OFF_THETA = 0;                  // [NW] floats; then, in place: dem_hi [NWB] bf16, dem_lo [NWB] bf16
    static constexpr int OFF_FO = NWB;                   // WU > 0: fm_out fp32 [NF], and the segment end states of the de-emphasis IIR
    static constexpr int OFF_ZS = OFF_FO + NF;
    static constexpr int LDS_FLOATS = WU > 0 ? OFF_ZS + NF / 8 + 16 : NWB;
    static constexpr int NCOL = NF / 16;
    static_assert(NF % 16 == 0 && NWB % 8 == 0 && OFF_FO % 4 == 0 && NWB >= NW + 1, "alignment");
};
// x = hi + lo (+ O(2^-16 x)), both halves as bf16 bit patterns in the upper 16 bits of a float
__device__ __forceinline__ void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    hi = f32_bits(x) & 0xffff0000u;
    lo = f32_bits(x - bits_f32(hi));
}
// bf16(a) low, bf16(b) high: one v_perm_b32 (bytes 2, 3 of a and bytes 2, 3 of b) instead of a shift and an and-or
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ float2 load_iq(const float2* p, unsigned i) {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p) + (size_t)(unsigned)(i * 8u));
}
__device__ __forceinline__ float2 load_iq(const uchar2* p, unsigned i) {
    const uchar2 v = *reinterpret_cast<const uchar2*>(reinterpret_cast<const char*>(p) + (size_t)(unsigned)(i * 2u));
    return make_float2((float)v.x - 127.0f, (float)v.y - 127.0f);  // reference src/app.cpp:56-62
}

// tolerance mode behind the first decimator: the stream is already the samples' phases in turns (k_predecim<.., true>)
__device__ __forceinline__ float2 load_iq(const float* p, unsigned i) {
    return make_float2(*reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + (size_t)(unsigned)(i * 4u)), 0.0f);
}

// two consecutive samples with one load
__device__ __forceinline__ float4 load_iq2(const float2* p, size_t i) { return *reinterpret_cast<const float4*>(p + i); }
__device__ __forceinline__ float4 load_iq2(const uchar2* p, size_t i) {
    const uchar4 v = *reinterpret_cast<const uchar4*>(p + i);
    return make_float4((float)v.x - 127.0f, (float)v.y - 127.0f, (float)v.z - 127.0f, (float)v.w - 127.0f);
}

// Stream-once traffic (the capture's samples, the audio frames): non-temporal accesses, so that the lines do not displace the planes the
// kernels hand each other through L2 / the Infinity Cache (fo_pl, pv_pl).  A/B switches of the development builds: -DFMD_NT_IN=0 / -DFMD_NT_OUT=0.
#ifndef FMD_NT_IN
#define FMD_NT_IN 1
#endif
#ifndef FMD_NT_OUT
#define FMD_NT_OUT 1
#endif
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#if FMD_NT_IN
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p)); return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
#if FMD_NT_IN
    const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p)); return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
// (A/B, -DFMD_NT_MID=1: the planes one kernel writes and the next reads ONCE — fm_out into the extract stage — with non-temporal loads)
#ifndef FMD_NT_MID
#define FMD_NT_MID 0
#endif
__device__ __forceinline__ float4 ld_mid(const float4* p) {
#if FMD_NT_MID
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p)); return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(float4* p, const float4& v) {
#if FMD_NT_OUT
    __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4*>(p));
#else
    *p = v;
#endif
}

template <typename InT, int TT = 512>
__global__ __launch_bounds__(256) void k_front(Dims d, const InT* __restrict__ in, const float2* __restrict__ tail_in,
                                               float2* __restrict__ tail_out, float2* __restrict__ fm_out_iq,
                                               float* __restrict__ fm_out_plain, float* __restrict__ fo_tail_out, FrontTaps taps,
                                               int deemph_path) {
    using G = FrontGeom<TT>;
    constexpr int T = G::T, NW = G::NW, NF = G::NF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* theta = smem + G::OFF_THETA;                           // [NW]
    float* dem = smem + G::OFF_DEM;                               // [NW-1]
    float* fo = smem + G::OFF_FO;                                 // [T+64]
    AtanTable* atab = reinterpret_cast<AtanTable*>(smem + G::OFF_ATAN);
    atan_table_fill(atab, threadIdx.x, 256);   // visible after the first barrier below

    const int tiles = d.n_fm_out / T;
    const int c = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles;
    const int o0 = tile * T;
    const int tid = threadIdx.x;
    const long g_lo = (long)2 * o0 - G::TAIL;                     // first input index of the tile (block relative)
    const InT* in_c = in + (size_t)c * d.N;
    const float2* tail_c = tail_in + (size_t)c * d.tail_base + (d.tail_base - G::TAIL);   // the newest TAIL of the tail_base samples kept

    // Input staging: every thread first issues ALL of its global loads (independent, 16 B per lane for cf32), then
    // consumes them — the memory-level parallelism is what keeps this kernel off the HBM-latency floor.
    {
        constexpr int PER = (NW + 255) / 256;
        float2 buf[PER];
        if (tile != 0) {
            // no history involved (workgroup-uniform): a uniform base and 32-bit offsets, so the loads take the SGPR-base form and
            // the address arithmetic stays out of the VALU (the general form below costs ~12 instructions per sample on 64-bit
            // adds, compares and selects: a tenth of this kernel)
            const unsigned g0 = (unsigned)g_lo;
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int i = tid + 256 * r;
                if (i < NW) buf[r] = load_iq(in_c, (unsigned)(g0 + (unsigned)i));
            }
        } else {
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int i = tid + 256 * r;
                if (i < NW) {
                    const int g = (int)g_lo + i;
                    buf[r] = (g < 0) ? tail_c[G::TAIL + g] : load_iq(in_c, (unsigned)g);
                }
            }
        }
        __syncthreads();   // the arctangent table
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int i = tid + 256 * r;
            if (i < NW) {
                theta[i] = fmd_atan2f_table<sizeof(InT) == 2>(buf[r].y, buf[r].x, atab);   // u8 IQ: small integers
            }
        }
    }
    __syncthreads();
    // a2: phase difference, wrap, scale — in place (dem aliases theta): every thread reads its pairs, then all write
    {
        const float pi = bits_f32(kPiBits), two_pi = bits_f32(kTwoPiBits);
        constexpr int PERD = (NW - 1 + 255) / 256;
        float dv[PERD];
#pragma unroll
        for (int r = 0; r < PERD; r++) {
            const int j = tid + 256 * r;
            if (j < NW - 1) {
                float dlt = theta[j + 1] - theta[j];
                if (dlt >= pi) dlt = dlt - two_pi;
                else if (dlt <= -pi) dlt = dlt + two_pi;
                dv[r] = dlt * taps.fm_gain;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PERD; r++) {
            const int j = tid + 256 * r;
            if (j < NW - 1) dem[j] = dv[r];
        }
    }
    __syncthreads();
    // a3: decimate-by-2 FIR, 8 lane accumulators.  Every thread makes TWO consecutive outputs from one window of 66 input
    // samples read as 17 ds_read_b128 (16 B lane stride: conflict-free, full LDS rate): 136 B of LDS traffic per output instead
    // of 256 B — and left to one output per thread the compiler pairs the 32 ds_read_b64 into ds_read2_b64, which move only
    // 128 B/clk.  This loop is the kernel's LDS hot spot.
    for (int pp = tid; 2 * pp < NF; pp += 256) {
        const int uu = 2 * pp;
        float w[68];
        const float4* src = reinterpret_cast<const float4*>(dem + 2 * uu);
#pragma unroll
        for (int q = 0; q < 17; q++) { const float4 t = src[q]; w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w; }
        float y[2];
#pragma unroll
        for (int v = 0; v < 2; v++) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 64; n++) acc[n & 7] = fmaf(w[2 * v + n], taps.b_fm_out[n], acc[n & 7]);
            const float a0 = acc[0] + acc[4], a1 = acc[1] + acc[5], a2 = acc[2] + acc[6], a3 = acc[3] + acc[7];
            y[v] = (a0 + a2) + (a1 + a3);
        }
        *reinterpret_cast<float2*>(fo + uu) = make_float2(y[0], y[1]);
        if (deemph_path && uu >= 64) *reinterpret_cast<float2*>(fm_out_plain + (size_t)c * d.n_fm_out + o0 + (uu - 64)) = make_float2(y[0], y[1]);
    }
    __syncthreads();
    // a5: Hilbert FIR; only lanes 1,3,5,7 of the reference's 8-lane accumulator see non-zero taps
    if (!deemph_path) {
        const float* fh = fo;                                  // the Hilbert FIR's window of output oo starts at fo[oo]
        for (int oo = tid; oo < T; oo += 256) {
            float l1 = 0.f, l3 = 0.f, l5 = 0.f, l7 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                l1 = fmaf(fh[oo + 1 + 8 * k], taps.b_hilbert_odd[4 * k + 0], l1);
                l3 = fmaf(fh[oo + 3 + 8 * k], taps.b_hilbert_odd[4 * k + 1], l3);
                l5 = fmaf(fh[oo + 5 + 8 * k], taps.b_hilbert_odd[4 * k + 2], l5);
                l7 = fmaf(fh[oo + 7 + 8 * k], taps.b_hilbert_odd[4 * k + 3], l7);
            }
            const float im = (0.0f + ((l1 + l5) + (l3 + l7))) + 0.0f;
            fm_out_iq[(size_t)c * d.n_fm_out + o0 + oo] = make_float2(fh[oo + 32], im);
        }
    }
    // keep the last TAIL input samples of the stream for the next block (and, when the de-emphasis path is
    // not the one maintaining it, the last 64 fm_out samples = the Hilbert FIR history the reference holds)
    if (tile == tiles - 1) {
        float2* tout = tail_out + (size_t)c * d.tail_base;
        for (int idx = tid; idx < d.tail_base; idx += 256) tout[idx] = load_iq(in_c, (unsigned)(d.N - d.tail_base + idx));
        if (!deemph_path && tid < 64) fo_tail_out[(size_t)c * 64 + tid] = fo[T + tid];
    }
}

// =============================================================================================
// k_front_mfma — FMD_FLAG_FAST_MATH form of k_front up to fm_out: the minimax arctangent (in turns), the decimating FIR as a
// bf16 x 3 matrix product (FrontGeomM above); WU > 0: with the de-emphasis IIR inside the tile (see k_front).  The Hilbert FIR is
// k_extract_bp's: the analytic signal never goes through HBM, only fm_out does (4 bytes per sample instead of 8).
// =============================================================================================
// The operands a front-end workgroup loads once: the Toeplitz images of the decimating FIR, and this lane's element of the operand that
// makes the pilot stage's four column sums from a tile of outputs (below): rows 0-3 = new.re, new.im, old.re, old.im weights (PllSparseTab)
struct FrontOps { bf16x8 adh[3], adl[3]; float4 wA; };
template <int WU>
__device__ __forceinline__ void load_front_ops(FrontOps& op, const uint4* __restrict__ tab, const PllSparseTab* __restrict__ sp, int lane, int lq) {
#pragma unroll
    for (int sK = 0; sK < 3; sK++) {
        op.adh[sK] = __builtin_bit_cast(bf16x8, tab[(sK * 2 + 0) * kWave + lane]);
        op.adl[sK] = __builtin_bit_cast(bf16x8, tab[(sK * 2 + 1) * kWave + lane]);
    }
    {
        const int r = lane & 15;                 // operand row: which of the four sums (rows 4-15: zero)
        const float* w = ((r & 1) ? sp->wim : sp->wre) + ((r & 2) ? 0 : 16) + 4 * lq;
        op.wA = r < 4 ? *reinterpret_cast<const float4*>(w) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

#ifdef FMD_F_PROBE
// development probe (tools/dbg/f_probe.py): cycles of the front end's workgroups between their barriers (wavefront 0 of every 61st workgroup)
__device__ unsigned long long g_f_probe[16];
#define F_STAMP(i_) do { const unsigned long long t_ = __builtin_readcyclecounter(); if ((blockIdx.x % 61) == 0 && threadIdx.x == 0) atomicAdd(&g_f_probe[i_], t_ - fp_t); fp_t = t_; } while (0)
#else
#define F_STAMP(i_)
#endif
// The front end from the tile's phases on (k_front_mfma, k_front_pre_mfma): theta[0 .. NW) = the phases (turns) of fm_in samples
// 2 o0 - TAIL ..., in LDS; dem32 = the 16-byte aligned start of that region, over which the discriminator output is written in place
// (theta may start up to 8 bytes into it).  All threads of the workgroup, phases complete (a barrier behind the writer).
// u8 captures at 256 kSa/s: the discriminator's wrap at EXACTLY half a turn.  Two consecutive integer samples in exactly opposite directions
// (a deep fade: a few LSB of signal), or a zero sample next to one on the negative real axis, differ by pi, and the reference's
//   if (d >= pi) d -= 2 pi; else if (d <= -pi) d += 2 pi                                   (fm_demod.cpp:36-43)
// then turns on the last bit of glibc's atan2f: its result is +pi or -pi, one full turn of the discriminator's range apart (a click of
// either sign).  The pair's outcome depends only on the first sample's direction: PllSparseTab::wrap_tie, one bit per u8 sample,
// made on the host with the exact restatement of atan2f (fmd_api.cpp).  Found by tests/test_gpu_realistic.py (fading captures).
__device__ __forceinline__ float wrap_tie_u8(float dflt, int x0, int y0, int x1, int y1, const uint32_t* __restrict__ tie) {
    const int cross = x1 * y0 - y1 * x0, dot = x1 * x0 + y1 * y0;
    if (cross == 0 && dot < 0) { const unsigned key = ((unsigned)(y0 + 127) << 8) | (unsigned)(x0 + 127); return ((tie[key >> 5] >> (key & 31u)) & 1u) ? 0.5f : -0.5f; }
    if ((x0 | y0) == 0 && y1 == 0 && x1 < 0) return -0.5f;       // atan2f(0, 0) = 0, atan2f(+0, x < 0) = +pi: a difference of +pi wraps to -pi
    if ((x1 | y1) == 0 && y0 == 0 && x0 < 0) return 0.5f;
    return dflt;
}
__device__ __forceinline__ int2 raw_u8(const uchar2* __restrict__ in_c, const float2* __restrict__ tail_c, int tail, int g) {
    if (g < 0) { const float2 v = tail_c[tail + g]; return make_int2((int)v.x, (int)v.y); }
    const uchar2 v = in_c[g];
    return make_int2((int)v.x - 127, (int)v.y - 127);
}

// DEM_READY (round 6, cf32 captures away from the block's start): the caller has written the discriminator output's bf16 halves itself — the phase
// difference as ONE arctangent of x[n] conj(x[n-1]), straight from the loaded samples (discriminator_pairs below) — and has passed a barrier.
template <int TT, int WU, bool TIES = false, bool DEM_READY = false>
__device__ __forceinline__ void front_from_phases(const Dims& d, float* smem, const float* theta, uint32_t* dem32, int c, int o0, int tid, float fm_gain,
                                                  const float* __restrict__ deemph, const FrontOps& op, float* __restrict__ fo_pl,
                                                  float4* __restrict__ pv_pl, const PllSparseTab* __restrict__ sp,
                                                  const uchar2* __restrict__ raw_c = nullptr, const float2* __restrict__ raw_tail = nullptr, int raw_tail_len = 0, int g_lo = 0) {
    using G = FrontGeomM<TT, WU>;
    constexpr int T = G::T, NW = G::NW, NF = G::NF;
    const int lane = tid & (kWave - 1), wv = tid >> 6, lrow = lane & 15, lq = lane >> 4;
    uint32_t* dem_hi32 = dem32;                                                    // two bf16 per word
    uint32_t* dem_lo32 = dem_hi32 + G::NWB / 2;
    float* fo = smem + G::OFF_FO;
    float4* pv_row = pv_pl + (size_t)c * (d.n_fm_out / 16) + o0 / 16;
#ifdef FMD_F_PROBE
    unsigned long long fp_t = __builtin_readcyclecounter();
#endif
    // phase difference, wrap, scale: two samples per thread and step, split into bf16 halves, in place over the phases
    if constexpr (!DEM_READY) {
        const float gain_t = fm_gain * bits_f32(kTwoPiBits);           // the discriminator's gain per turn
        constexpr int NPW = G::NWB / 2;                              // words per half
        constexpr int PERP = (NPW + 255) / 256;
        uint32_t wh[PERP], wl[PERP];
#pragma unroll
        for (int r = 0; r < PERP; r++) {
            const int pw = tid + 256 * r, j = 2 * pw;
            wh[r] = 0u; wl[r] = 0u;
            if (j < NW - 1) {
                const float t0 = theta[j], t1 = theta[j + 1], t2 = (j + 2 < NW) ? theta[j + 2] : t1;
                float d0 = t1 - t0, d1 = t2 - t1;
                d0 = d0 - rintf(d0); d1 = d1 - rintf(d1);           // reference fm_demod.cpp:36-43: the phase difference wrapped to half a turn
                if constexpr (TIES) {
                    if (__builtin_expect(fabsf(fabsf(d0) - 0.5f) < 2.0e-6f || fabsf(fabsf(d1) - 0.5f) < 2.0e-6f, 0)) {      // (wrap_tie_u8 above)
                        const int2 s0 = raw_u8(raw_c, raw_tail, raw_tail_len, g_lo + j), s1 = raw_u8(raw_c, raw_tail, raw_tail_len, g_lo + j + 1);
                        if (fabsf(fabsf(d0) - 0.5f) < 2.0e-6f) d0 = wrap_tie_u8(d0, s0.x, s0.y, s1.x, s1.y, sp->wrap_tie);
                        if (j + 2 < NW && fabsf(fabsf(d1) - 0.5f) < 2.0e-6f) {
                            const int2 s2 = raw_u8(raw_c, raw_tail, raw_tail_len, g_lo + j + 2);
                            d1 = wrap_tie_u8(d1, s1.x, s1.y, s2.x, s2.y, sp->wrap_tie);
                        }
                    }
                }
                d0 *= gain_t; d1 = (j + 1 < NW - 1) ? d1 * gain_t : 0.0f;
                uint32_t h0, l0, h1, l1;
                split_bf16(d0, h0, l0); split_bf16(d1, h1, l1);
                wh[r] = pack_hi16(h0, h1); wl[r] = pack_hi16(l0, l1);
            }
        }
        F_STAMP(2);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PERP; r++) {
            const int pw = tid + 256 * r;
            if (pw < NPW) { dem_hi32[pw] = wh[r]; dem_lo32[pw] = wl[r]; }
        }
    }
    F_STAMP(3);
    if constexpr (!DEM_READY) __syncthreads();
    F_STAMP(4);
    // a3: decimate-by-2 FIR: wavefront w takes the 16-column tiles w, w + 4, ... (a column = 16 consecutive outputs).
    // fm_out goes to its plane undelayed (the rows start with the previous block's tail, k_pll_sparse); the consumers delay it by 32
    // for the real rail and k_extract_bp's composite FIRs contain the Hilbert rail.
    float* fo_row = fo_pl + (size_t)c * (kFoPad + d.n_fm_out) + kFoPad + o0;
    for (int ct = wv; ct * 16 < G::NCOL; ct += 4) {
        const int col = ct * 16 + lrow, colr = col < G::NCOL ? col : G::NCOL - 1;
        // (round 3, PMC: a wavefront of this kernel spent 37 % of its cycles waiting for the previous MFMA of one nine-long chain)
        f32x4 acc, acc1, acc2;   // three chains (hi hi, lo hi, hi lo) instead of one three times as long
#pragma unroll
        for (int sK = 0; sK < 3; sK++) {
            const int e = 32 * colr + 32 * sK + 8 * lq;              // bf16 element index, a multiple of 8
            const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(dem_hi32 + e / 2));
            const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(dem_lo32 + e / 2));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op.adh[sK], bh, sK ? acc : kZero4, 0, 0, 0);      // (the first step starts from the constant 0: no registers to clear)
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op.adl[sK], bh, sK ? acc1 : kZero4, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op.adh[sK], bl, sK ? acc2 : kZero4, 0, 0, 0);
        }
        acc = acc + (acc1 + acc2);
        if constexpr (WU == 0) {
            // the pilot stage's column sums (k_pll_sparse then reads 16 bytes per column instead of fm_out's 64): S = W Y, W = the four weight
            // rows (16 x 16, rows 4-15 zero), Y = this tile of fp32 outputs as it lies in the accumulators — four v_mfma_f32_16x16x4_f32
            // steps, step j taking output rows 4 k + j (lane (column, k) holds exactly acc[j]); the four sums of a column land in ONE lane.
            // (Was 16 fp32 FMAs per lane and two lane-swap reductions: 26 VALU instructions a tile and the swaps' latency on a kernel that is
            // HBM- and VALU-limited together; the matrix pipe is idle here.  fp32 operands and accumulation.)
            f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x4f32(op.wA.x, acc[0], kZero4, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_16x16x4f32(op.wA.y, acc[1], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_16x16x4f32(op.wA.z, acc[2], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_16x16x4f32(op.wA.w, acc[3], sv, 0, 0, 0);
            if (col < G::NCOL && lq == 0) pv_row[col] = make_float4(sv[0], sv[1], sv[2], sv[3]);       // 16 lanes, 256 bytes in a row
        }
        if (col < G::NCOL) {
            if constexpr (WU == 0) *reinterpret_cast<float4*>(fo_row + 16 * col + 4 * lq) = make_float4(acc[0], acc[1], acc[2], acc[3]);   // a wavefront: 4 KB in a row
            else *reinterpret_cast<float4*>(fo + 16 * col + 4 * lq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    F_STAMP(5);
    if constexpr (WU > 0) {
        __syncthreads();
        // a4 in the tile (see k_front): 8 samples per thread from a zero state, end states through LDS, 16 segments of history
        const float b0 = deemph[4 * c + 0], b1 = deemph[4 * c + 1], a0 = deemph[4 * c + 2];
        if (deemph[4 * c + 3] != 0.0f) {
            float* zs = smem + G::OFF_ZS;
            constexpr int NSEG = NF / 8;
            float yv[8];
            if (tid < 16) zs[tid] = 0.0f;
            if (tid < NSEG) {
                const float4 xa = *reinterpret_cast<const float4*>(fo + 8 * tid), xb = *reinterpret_cast<const float4*>(fo + 8 * tid + 4);
                const float xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                float xp = tid ? fo[8 * tid - 1] : 0.0f, z = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; k++) { z = fmaf(a0, z, fmaf(xs[k], b1, xp * b0)); yv[k] = z; xp = xs[k]; }
                zs[16 + tid] = z;
            }
            __syncthreads();
            if (tid < NSEG) {
                const float a2 = a0 * a0, a4 = a2 * a2, a8 = a4 * a4;
                float e = 0.0f;
#pragma unroll
                for (int k = 0; k < 16; k++) e = fmaf(a8, e, zs[tid + k]);
                float pw = a0;
#pragma unroll
                for (int k = 0; k < 8; k++) { yv[k] = fmaf(pw, e, yv[k]); pw *= a0; }
                *reinterpret_cast<float4*>(fo + 8 * tid) = make_float4(yv[0], yv[1], yv[2], yv[3]);
                *reinterpret_cast<float4*>(fo + 8 * tid + 4) = make_float4(yv[4], yv[5], yv[6], yv[7]);
            }
            __syncthreads();
        }
        for (int q4 = tid; q4 < T / 4; q4 += 256) *reinterpret_cast<float4*>(fo_row + 4 * q4) = *reinterpret_cast<const float4*>(fo + WU + 4 * q4);
        // the pilot stage's column sums from the de-emphasised outputs (the reference filters fm_out in place ahead of every consumer,
        // broadcast_fm_demod.cpp:403-406): the same fp32 matrix product as above, the tile of outputs read back from LDS in the
        // accumulators' layout (lane (column, k): outputs 4 k .. 4 k + 3 of the column)
        for (int ct = wv; ct * 16 < T / 16; ct += 4) {
            const int col = ct * 16 + lrow;
            const float4 y = *reinterpret_cast<const float4*>(fo + WU + 16 * col + 4 * lq);
            f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x4f32(op.wA.x, y.x, kZero4, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_16x16x4f32(op.wA.y, y.y, sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_16x16x4f32(op.wA.z, y.z, sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_16x16x4f32(op.wA.w, y.w, sv, 0, 0, 0);
            if (lq == 0) pv_row[col] = make_float4(sv[0], sv[1], sv[2], sv[3]);
        }
    }
}

// FUSED: the launch's first pf.n_wg workgroups run the pilot stage (k_pll_sparse's body) of the block BEFORE this one — on the deferred
// schedule (fmd_api.cpp) the front end's queue then carries front(k + 1) + pilot(k), extract(k), front(k + 2) + pilot(k + 1), ... and no
// kernel on it waits for another queue; the pilot stage's half-thousand wavefronts start first and finish inside the front end's HBM-bound time.
template <typename InT, int TT, int WU, bool FUSED = false>
__global__ __launch_bounds__(256, FUSED ? 6 : 5) void k_front_mfma(Dims d, const InT* __restrict__ in, const float2* __restrict__ tail_in,
                                                    float2* __restrict__ tail_out, float* __restrict__ fo_pl, float fm_gain,
                                                    const float* __restrict__ deemph, const uint4* __restrict__ tab, float4* __restrict__ pv_pl,
                                                    const PllSparseTab* __restrict__ sp, PllFusedArgs pf) {
    using G = FrontGeomM<TT, WU>;
    constexpr int T = G::T, NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = (int)blockIdx.x;
    if constexpr (FUSED) {
        if (bid < pf.n_wg) {
            pll_sparse_body(d, bid * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & (kWave - 1)), pf.pv, pf.hist_in, pf.hist_out, pf.fo, pf.fo_next, pf.poly, pf.poly_next,
                            pf.state, pf.k, pf.tab, 0, pf.spec_stats);
            return;
        }
        bid -= pf.n_wg;
    }
    float* theta = smem + G::OFF_THETA;

    const int tiles = d.n_fm_out / T;
    const int c = bid / tiles, tile = bid % tiles, o0 = tile * T, tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < 256);     // (the bounds of the unrolled staging loops are tested against it)
    const int lane = tid & (kWave - 1), lq = lane >> 4;
    const int g_lo = 2 * o0 - G::TAIL;                            // first input index of the tile (block relative)
    const InT* in_c = in + (size_t)c * d.N;
    const float2* tail_c = tail_in + (size_t)c * d.tail_base + (d.tail_base - G::TAIL);

    FrontOps op;                                                   // (with the other early loads)
    load_front_ops<WU>(op, tab, sp, lane, lq);
#ifdef FMD_F_PROBE
    unsigned long long fp_t = __builtin_readcyclecounter();
    if ((blockIdx.x % 61) == 0 && threadIdx.x == 0) atomicAdd(&g_f_probe[7], 1ull);
#endif
    // a0-a2: staging, arctangent (all loads first)
    bool staged = false;
// Round 6 A/B (profiles/round6/front_conj_ab.txt): the cf32 discriminator as one arctangent of x[n] conj(x[n-1]) straight from
// the loaded samples (no phase array in LDS, one barrier in the tile instead of three) against round 5's two arctangents through LDS: same
// results to 1e-6, and SLOWER — k_front_mfma 0.163 ms against 0.152 in the step (4096 stations), 0.348 against 0.32 at 8192: the 8-byte load
// of the sample in front of every pair is one more vector-memory instruction per two samples on a kernel that already keeps the memory pipe
// and the vector ALU busy together, and what it saves (LDS round trips, two barriers) was hidden by the six workgroups a CU holds.
// Kept behind the macro; k_chain (fmd_kernels_chain.inc) uses the form where it has no LDS for a phase array.
#ifndef FMD_FRONT_CONJ
#define FMD_FRONT_CONJ 0
#endif
#if FMD_FRONT_CONJ
    if constexpr (sizeof(InT) == 8) {
        // cf32 input away from the block's start (round 6): the discriminator's phase difference as ONE arctangent of
        // x[n] conj(x[n-1]) — the reference takes the two samples' arctangents and wraps their difference (fm_demod.cpp:30-45): the same angle,
        // in (-1/2, 1/2] turns by construction — straight from the loaded samples: word w of the output (elements 2 w, 2 w + 1) needs the
        // tile's samples 2 w, 2 w + 1, 2 w + 2 = the 16-byte pair at the even block sample g_lo + 2 w + 1 and the 8 bytes in front of it.  No
        // phase array in LDS (one write and 1.5 reads per sample), no wrap, and one barrier in the tile instead of three.
        if (tile != 0) {
            constexpr int NWORD = (NW - 1 + 1) / 2;                       // 1055 words hold the NW - 1 differences (the last word's second element is 0)
            constexpr int PERW = NWORD / 256, RESTW = NWORD - 256 * PERW;  // (1024-sample tiles: four full rounds and 31 words for the first lanes)
            const float gain_t = fm_gain * bits_f32(kTwoPiBits);
            uint32_t* dem_hi32 = reinterpret_cast<uint32_t*>(smem);
            uint32_t* dem_lo32 = dem_hi32 + G::NWB / 2;
            const float2* base = in_c + (g_lo + 1);                       // (even sample: 16-byte aligned)
            float4 qb[PERW + 1]; float2 pb[PERW + 1];
#pragma unroll
            for (int r = 0; r < PERW; r++) { const int w = tid + 256 * r; qb[r] = ld_stream(reinterpret_cast<const float4*>(base + 2 * w)); pb[r] = base[2 * w - 1]; }
            if (tid < RESTW) { const int w = 256 * PERW + tid; qb[PERW] = ld_stream(reinterpret_cast<const float4*>(base + 2 * w)); pb[PERW] = base[2 * w - 1]; }
            auto pair = [&](const float4& q, const float2& p, int w) __attribute__((always_inline)) {
                const float re0 = fmaf(q.x, p.x, q.y * p.y), im0 = fmaf(q.y, p.x, -(q.x * p.y));        // x[2w+1] conj(x[2w])
                const float re1 = fmaf(q.z, q.x, q.w * q.y), im1 = fmaf(q.w, q.x, -(q.z * q.y));        // x[2w+2] conj(x[2w+1])
                const float d0 = fast_atan2_turns(im0, re0) * gain_t;
                const float d1 = (2 * w + 1 < NW - 1) ? fast_atan2_turns(im1, re1) * gain_t : 0.0f;
                uint32_t h0, l0, h1, l1;
                split_bf16(d0, h0, l0); split_bf16(d1, h1, l1);
                dem_hi32[w] = pack_hi16(h0, h1); dem_lo32[w] = pack_hi16(l0, l1);
            };
#pragma unroll
            for (int r = 0; r < PERW; r++) pair(qb[r], pb[r], tid + 256 * r);
            if (tid < RESTW) pair(qb[PERW], pb[PERW], 256 * PERW + tid);
            else if (256 * PERW + tid < G::NWB / 2) { dem_hi32[256 * PERW + tid] = 0u; dem_lo32[256 * PERW + tid] = 0u; }      // (the padding under the zero taps)
            F_STAMP(0);
            __syncthreads();
            F_STAMP(1);
            front_from_phases<TT, WU, false, true>(d, smem, theta, reinterpret_cast<uint32_t*>(smem), c, o0, tid, fm_gain, deemph, op, fo_pl, pv_pl, sp);
            if (tile == tiles - 1) {
                float2* tout = tail_out + (size_t)c * d.tail_base;
                for (int idx = tid; idx < d.tail_base; idx += 256) tout[idx] = load_iq(in_c, (unsigned)(d.N - d.tail_base + idx));
            }
            return;
        }
    }
#else      // (round 5's form, kept for the A/B: tools/build_variant.sh name "-DFMD_FRONT_CONJ=0")
    if constexpr (sizeof(InT) == 8) {
        // cf32 input away from the block's start: 16 bytes per lane (two samples).  The tile starts at an odd sample (TAIL is odd): the
        // pairs start one sample earlier, pair q = samples 2 q - 1 and 2 q of the tile; what is left behind the full rounds, one per thread.
        if (tile != 0) {
            constexpr int NQF = ((NW + 1) / 2) / 256 * 256, PERQ = NQF / 256, REST0 = 2 * NQF - 1, PERR = (NW - REST0 + 255) / 256;
            const float4* src = reinterpret_cast<const float4*>(in_c + (g_lo - 1));
            float4 qb[PERQ]; float2 rb[PERR];
#pragma unroll
            for (int r = 0; r < PERQ; r++) qb[r] = ld_stream(src + tid + 256 * r);
#pragma unroll
            for (int r = 0; r < PERR; r++) { const int i = REST0 + tid + 256 * r; if (i < NW) rb[r] = load_iq(in_c, (unsigned)(g_lo + i)); }
#pragma unroll
            for (int r = 0; r < PERQ; r++) {
                const int q = tid + 256 * r;
                const float t1 = fast_atan2_turns(qb[r].w, qb[r].z);
                if (q > 0) theta[2 * q - 1] = fast_atan2_turns(qb[r].y, qb[r].x);
                theta[2 * q] = t1;
            }
#pragma unroll
            for (int r = 0; r < PERR; r++) { const int i = REST0 + tid + 256 * r; if (i < NW) theta[i] = fast_atan2_turns(rb[r].y, rb[r].x); }
            staged = true;
        }
    }
#endif
    if constexpr (sizeof(InT) == 2 && (NW + 1) / 8 >= 256) {
        // u8 input away from the block's start: 16 bytes per lane are eight samples (the tile starts at sample 1 of such a group:
        // octet o = samples 8 o - 1 .. 8 o + 6 of the tile); one full round, the rest one sample per thread.
        if (tile != 0) {
            constexpr int NOF = ((NW + 1) / 8) / 256 * 256, PERO = NOF / 256, REST0 = 8 * NOF - 1, PERR = (NW - REST0 + 255) / 256;
            static_assert((G::TAIL & 7) == 7, "the tile starts at sample 1 (mod 8) of the row");
            const uint4* src = reinterpret_cast<const uint4*>(in_c + (g_lo - 1));
            uint4 ob[PERO]; float2 rb[PERR];
#pragma unroll
            for (int r = 0; r < PERO; r++) ob[r] = ld_stream(src + tid + 256 * r);
#pragma unroll
            for (int r = 0; r < PERR; r++) { const int i = REST0 + tid + 256 * r; if (i < NW) rb[r] = load_iq(in_c, (unsigned)(g_lo + i)); }
#pragma unroll
            for (int r = 0; r < PERO; r++) {
                const int o = tid + 256 * r;
                const uint32_t w[4] = {ob[r].x, ob[r].y, ob[r].z, ob[r].w};
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t ww = w[u >> 1] >> (16 * (u & 1));
                    const float re = (float)(ww & 0xffu) - 127.0f, im = (float)((ww >> 8) & 0xffu) - 127.0f;   // reference src/app.cpp:56-62
                    const float t = fast_atan2_turns(im, re);
                    if (u > 0 || o > 0) theta[8 * o - 1 + u] = t;
                }
            }
#pragma unroll
            for (int r = 0; r < PERR; r++) { const int i = REST0 + tid + 256 * r; if (i < NW) theta[i] = fast_atan2_turns(rb[r].y, rb[r].x); }
            staged = true;
        }
    }
    if (!staged) {
        constexpr int PER = (NW + 255) / 256;
        float2 buf[PER];
        if (tile != 0) {
            const unsigned g0 = (unsigned)g_lo;
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int i = tid + 256 * r;
                if (i < NW) buf[r] = load_iq(in_c, (unsigned)(g0 + (unsigned)i));
            }
        } else {
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int i = tid + 256 * r;
                if (i < NW) {
                    const int g = g_lo + i;
                    buf[r] = (g < 0) ? tail_c[G::TAIL + g] : load_iq(in_c, (unsigned)g);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int i = tid + 256 * r;
            if (i < NW) {   // phases in turns: the wrap below is x - rint(x)
                if constexpr (sizeof(InT) == 4) theta[i] = buf[r].x;      // (1.024 / 2.048 MSa/s: k_predecim has taken the arctangent)
                else theta[i] = fast_atan2_turns(buf[r].y, buf[r].x);
            }
        }
    }
    F_STAMP(0);
    __syncthreads();
    F_STAMP(1);
    if constexpr (sizeof(InT) == 2) front_from_phases<TT, WU, true>(d, smem, theta, reinterpret_cast<uint32_t*>(smem), c, o0, tid, fm_gain, deemph, op, fo_pl, pv_pl, sp,
                                                                    in_c, tail_c, G::TAIL, g_lo);
    else front_from_phases<TT, WU>(d, smem, theta, reinterpret_cast<uint32_t*>(smem), c, o0, tid, fm_gain, deemph, op, fo_pl, pv_pl, sp);
    if (tile == tiles - 1) {
        float2* tout = tail_out + (size_t)c * d.tail_base;
        for (int idx = tid; idx < d.tail_base; idx += 256) tout[idx] = load_iq(in_c, (unsigned)(d.N - d.tail_base + idx));
    }
}

// =============================================================================================
// k_predecim — the first decimator on its own at 1.024 / 2.048 MSa/s: PolyphaseDownsampler<cf32> M x 64 taps
// (polyphase_filter.h:41-64, c32_f32_cum_mul.cpp:70-111) from baseband (cf32 or u8, reference src/app.cpp:56-62) to the
// 256 kSa/s stream fm_in, which k_front then takes exactly as it takes a 256 kSa/s capture.
// Fused into one kernel with the discriminator (as it was at first) the decimator had to be recomputed over the 191-sample
// halo of the FIRs behind it (+19 % of the dominant work at 512-output tiles, which is all its 52 KB of LDS allowed); on its own
// the halo is 64 - M input samples per 512 outputs, every lane is busy in the one pair-blocked pass, and the back half runs
// on the 1024-output tiles of the 256 kSa/s path.  fm_in costs 16 B per 256 kSa/s sample of extra HBM traffic (written once,
// read once) on a path that sits at a fifth of the HBM roofline.
// One workgroup = one channel x 512 fm_in outputs; input split into M phases in LDS; accumulation order as in k_front.
// =============================================================================================
template <int M>
struct PredecimGeom {
    static constexpr int TP = 512;                    // outputs per workgroup
    static constexpr int NJ = 64 / M;                 // taps per phase
    static constexpr int NB = M * TP + 64;            // input samples staged (64 - M of history, M TP of the tile, M of overhang)
    static constexpr int Q = NB / M;                  // entries per phase
    static constexpr int PSR = 16 / M;
    static constexpr int PS = ((Q - PSR + 15) / 16) * 16 + PSR;
    static constexpr int HIST = 64;                   // input samples of history kept per channel (64 - M are used)
};

// THETA (tolerance mode): the discriminator's arctangent is taken here and fm_in is the samples' PHASES in turns, 4 bytes instead
// of 8 — this kernel is bandwidth-bound on cf32 input and has the VALU to spare, and the round trip of fm_in through HBM halves
template <int M, typename InT, bool THETA = false>
__global__ __launch_bounds__(256) void k_predecim(Dims d, const InT* __restrict__ in, const float2* __restrict__ tail_in,
                                                  float2* __restrict__ tail_out, float2* __restrict__ fm_in, FrontTaps taps) {
    using G = PredecimGeom<M>;
    constexpr int TP = G::TP, NJ = G::NJ, NB = G::NB, PS = G::PS;
    __shared__ __attribute__((aligned(16))) float2 ph[M * PS];
    const int tiles = d.n_fm_in / TP;
    const int c = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int n0 = tile * TP, tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < 256);     // (the bounds of the unrolled staging loops are tested against it)
    const long g_lo = (long)M * n0 + M - 64;          // first input sample of the tile's first output (block relative)
    const InT* in_c = in + (size_t)c * d.N;
    const float2* tail_c = tail_in + (size_t)c * G::HIST;
    {
        constexpr int ITEMS = NB / 2, PER = (ITEMS + 255) / 256;   // items of 2 consecutive samples (g_lo is even)
        float4 buf[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int j = tid + 256 * r;
            if (j < ITEMS) {
                long g = g_lo + 2 * j;
                g = g < d.N - 2 ? g : d.N - 2;                       // the last tile's overhang: read, never used
                buf[r] = (g < 0) ? *reinterpret_cast<const float4*>(tail_c + (G::HIST + g)) : load_iq2(in_c, (size_t)g);
            }
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int j = tid + 256 * r;
            if (j < ITEMS) {
                const int i0 = 2 * j, i1 = 2 * j + 1;
                ph[(i0 % M) * PS + (i0 / M)] = make_float2(buf[r].x, buf[r].y);
                ph[(i1 % M) * PS + (i1 / M)] = make_float2(buf[r].z, buf[r].w);
            }
        }
    }
    __syncthreads();
    {
        // two consecutive outputs per thread from one sliding window per phase (see k_front); lane (n & 3) sums taps n in
        // increasing n, then (l0+l2)+(l1+l3)
        constexpr int NV = (NJ + 2) / 2;
        const int il = 2 * tid;
        float ar[2][4], ai[2][4];
#pragma unroll
        for (int v = 0; v < 2; v++)
#pragma unroll
            for (int q = 0; q < 4; q++) { ar[v][q] = 0.f; ai[v][q] = 0.f; }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            float2 w[M / 4][2 * NV];
#pragma unroll
            for (int h = 0; h < M / 4; h++) {
                const float4* src = reinterpret_cast<const float4*>(ph + (p + 4 * h) * PS + il);
#pragma unroll
                for (int k = 0; k < NV; k++) { const float4 t = src[k]; w[h][2 * k] = make_float2(t.x, t.y); w[h][2 * k + 1] = make_float2(t.z, t.w); }
            }
#pragma unroll
            for (int jj = 0; jj < NJ; jj++) {
#pragma unroll
                for (int h = 0; h < M / 4; h++) {
                    const float b = taps.b_fm_in[M * jj + p + 4 * h];
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        ar[v][p] = fmaf(w[h][jj + v].x, b, ar[v][p]);
                        ai[v][p] = fmaf(w[h][jj + v].y, b, ai[v][p]);
                    }
                }
            }
        }
        float4 o;
        o.x = (ar[0][0] + ar[0][2]) + (ar[0][1] + ar[0][3]); o.y = (ai[0][0] + ai[0][2]) + (ai[0][1] + ai[0][3]);
        o.z = (ar[1][0] + ar[1][2]) + (ar[1][1] + ar[1][3]); o.w = (ai[1][0] + ai[1][2]) + (ai[1][1] + ai[1][3]);
        if constexpr (THETA)
            *reinterpret_cast<float2*>(reinterpret_cast<float*>(fm_in) + (size_t)c * d.n_fm_in + n0 + il) = make_float2(fast_atan2_turns(o.y, o.x), fast_atan2_turns(o.w, o.z));
        else
            *reinterpret_cast<float4*>(fm_in + (size_t)c * d.n_fm_in + n0 + il) = o;
    }
    // the last 64 input samples of the block are the next block's history
    if (tile == tiles - 1 && tid < G::HIST) tail_out[(size_t)c * G::HIST + tid] = load_iq(in_c, (unsigned)(d.N - G::HIST + tid));
}

// =============================================================================================
// k_predecim_mfma — FMD_FLAG_FAST_MATH form of k_predecim: the decimate-by-M FIR of both rails as bf16 x 3 matrix products
// (the scheme of FrontGeomM), the discriminator's arctangent taken on the accumulators; fm_in = phases in turns.
//   Y[m][col] = y[16 col + m] = sum_t A[m][t] w[16 M col + t],   A[m][t] = b[t - SH - M m]   (t < 64 + SH + 15 M: 4 / 6 K-steps of 32)
// k_predecim spends 128 VALU FMAs per output and is VALU-bound on u8 input and co-limited on cf32; here the VALU only splits the
// samples into bf16 halves (u8 samples ARE bf16 numbers: one half, two MFMAs per K-step instead of three) and the kernel streams.
// LDS: one bf16 array per rail and half, 8 elements of padding per column stride (16 M elements) so that the 16 lanes of a
// ds_read_b128 pass (one per column) hit 16 different 16-byte bank groups.
// =============================================================================================
template <int M, int TPX>
struct PredecimGeomM {
    static constexpr int TP = TPX;                    // outputs per workgroup
    static constexpr int NT = TP / 256;               // 16 x 16 output tiles
    static constexpr int SH = 8 - M;                  // the staged window starts SH samples early, on a multiple of 8 samples: 16-byte loads of u8 captures
    static constexpr int NB = M * TP + 64;            // input samples staged: w[j] = s[M n0 + M - 64 - SH + j];  A[m][t] = b[t - SH - M m]
    static constexpr int KS = (64 + SH + 15 * M + 31) / 32;
    static constexpr int SEG = 16 * M;                // elements between two columns' windows
    static constexpr int NEP = NB + 8 * (NB / SEG + 1);   // padded elements per array
    static constexpr int HIST = 64;
    static_assert(NB % 8 == 0 && SEG * (TP / 16 - 1) + 32 * KS <= NB, "every operand read stays inside the staged samples");
};
// x = hi + lo + O(2^-17 x): both halves rounded (half up) to bf16, as bit patterns in the upper 16 bits
__device__ __forceinline__ void split_bf16_rn(float x, uint32_t& hi, uint32_t& lo) {
    hi = (f32_bits(x) + 0x8000u) & 0xffff0000u;
    lo = f32_bits(x - bits_f32(hi)) + 0x8000u;
}
template <int M, typename InT, int TPX>
__global__ __launch_bounds__(256) void k_predecim_mfma(Dims d, const InT* __restrict__ in, const float2* __restrict__ tail_in,
                                                       float2* __restrict__ tail_out, float* __restrict__ fm_in, const uint4* __restrict__ tab) {
    using G = PredecimGeomM<M, TPX>;
    constexpr int TP = G::TP, NB = G::NB, KS = G::KS, SEG = G::SEG;
    constexpr bool U8 = sizeof(InT) == 2;
    constexpr int NH = U8 ? 1 : 2;                                    // halves kept per rail
    constexpr int NWD = G::NEP / 2;                                   // words per array
    __shared__ __attribute__((aligned(16))) uint32_t arr[2 * NH * NWD];   // [rail][half]
    const int tiles = d.n_fm_in / TP;
    const int c = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int n0 = tile * TP, tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < 256);
    const int lane = tid & (kWave - 1), wv = tid >> 6, lrow = lane & 15, lq = lane >> 4;
    const int g_lo = M * n0 + M - 64 - G::SH;                         // first input sample staged (block relative), a multiple of 8
    const InT* in_c = in + (size_t)c * d.N;
    const float2* tail_c = tail_in + (size_t)c * G::HIST;
    auto word_of = [](int pair) { return pair + 4 * (pair / (SEG / 2)); };      // pair p = elements 2 p, 2 p + 1 -> word of the padded array
    // the Toeplitz operands (cf32: not before the staging, whose load buffers are the register peak; u8: first, the staging is short)
    bf16x8 adh[KS], adl[KS];
    auto load_operands = [&]() {
#pragma unroll
        for (int sK = 0; sK < KS; sK++) {
            adh[sK] = __builtin_bit_cast(bf16x8, tab[(sK * 2 + 0) * kWave + lane]);
            adl[sK] = __builtin_bit_cast(bf16x8, tab[(sK * 2 + 1) * kWave + lane]);
        }
    };
    if constexpr (U8) load_operands();
    if constexpr (!U8) {
        constexpr int ITEMS = NB / 2, PER = (ITEMS + 255) / 256;      // two samples (16 bytes) per lane and round; all loads first
        float4 buf[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int j = tid + 256 * r;
            if (j < ITEMS) {
                int g = g_lo + 2 * j;
                g = g < d.N - 2 ? g : d.N - 2;                       // the last tile's overhang (M samples under zero taps): read inside the row
                buf[r] = (g < 0) ? *reinterpret_cast<const float4*>(tail_c + (G::HIST + g)) : *reinterpret_cast<const float4*>(in_c + g);
            }
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int j = tid + 256 * r;
            if (j < ITEMS) {
                uint32_t h0, l0, h1, l1;
                const int w = word_of(j);
                split_bf16_rn(buf[r].x, h0, l0); split_bf16_rn(buf[r].z, h1, l1);
                arr[0 * NWD + w] = pack_hi16(h0, h1); arr[1 * NWD + w] = pack_hi16(l0, l1);
                split_bf16_rn(buf[r].y, h0, l0); split_bf16_rn(buf[r].w, h1, l1);
                arr[2 * NWD + w] = pack_hi16(h0, h1); arr[3 * NWD + w] = pack_hi16(l0, l1);
            }
        }
    } else {
        constexpr int ITEMS = NB / 8, PER = (ITEMS + 255) / 256;      // eight samples (16 bytes) per lane and round
        uint4 buf[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int j = tid + 256 * r, g = g_lo + 8 * j;
            if (j < ITEMS && g >= 0) buf[r] = *reinterpret_cast<const uint4*>(in_c + (g < d.N - 8 ? g : d.N - 8));      // (the last tile's overhang: inside the row)
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int j = tid + 256 * r, g = g_lo + 8 * j;
            if (j < ITEMS) {
                uint32_t re[8], im[8];
                if (g < 0) {        // the block's first tile: history, kept as cf32 (u8 captures: integers; after a cf32 block: rounded to bf16)
#pragma unroll
                    for (int u = 0; u < 8; u++) { const float2 v = tail_c[G::HIST + g + u]; re[u] = f32_bits(v.x) + 0x8000u; im[u] = f32_bits(v.y) + 0x8000u; }
                } else {
                    const uint32_t w4[4] = {buf[r].x, buf[r].y, buf[r].z, buf[r].w};
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t ww = w4[u >> 1] >> (16 * (u & 1));
                        re[u] = f32_bits((float)(ww & 0xffu) - 127.0f); im[u] = f32_bits((float)((ww >> 8) & 0xffu) - 127.0f);      // reference src/app.cpp:56-62
                    }
                }
                // (integers of at most 8 bits: exact in bf16)
                const int w = word_of(4 * j);
                *reinterpret_cast<uint4*>(arr + 0 * NWD + w) = make_uint4(pack_hi16(re[0], re[1]), pack_hi16(re[2], re[3]), pack_hi16(re[4], re[5]), pack_hi16(re[6], re[7]));
                *reinterpret_cast<uint4*>(arr + 1 * NWD + w) = make_uint4(pack_hi16(im[0], im[1]), pack_hi16(im[2], im[3]), pack_hi16(im[4], im[5]), pack_hi16(im[6], im[7]));
            }
        }
    }
    if constexpr (!U8) { if (wv < G::NT) load_operands(); }
    __syncthreads();
    float* out_c = fm_in + (size_t)c * d.n_fm_in + n0;
    for (int ct = wv; ct < G::NT; ct += 4) {
        const int col = 16 * ct + lrow;
        f32x4 acc[2][3];
#pragma unroll
        for (int sK = 0; sK < KS; sK++) {
            const int e = SEG * col + 32 * sK + 8 * lq;              // element index, a multiple of 8
            const int w = (e + 8 * (e / SEG)) / 2;
#pragma unroll
            for (int rail = 0; rail < 2; rail++) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arr + (rail * NH) * NWD + w));
                acc[rail][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adh[sK], bh, sK ? acc[rail][0] : kZero4, 0, 0, 0);
                acc[rail][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adl[sK], bh, sK ? acc[rail][1] : kZero4, 0, 0, 0);
                if constexpr (!U8) {
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arr + (rail * NH + 1) * NWD + w));
                    acc[rail][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adh[sK], bl, sK ? acc[rail][2] : kZero4, 0, 0, 0);
                }
            }
        }
        f32x4 yr, yi;
        if constexpr (U8) { yr = acc[0][0] + acc[0][1]; yi = acc[1][0] + acc[1][1]; }
        else { yr = acc[0][0] + (acc[0][1] + acc[0][2]); yi = acc[1][0] + (acc[1][1] + acc[1][2]); }
        *reinterpret_cast<float4*>(out_c + 16 * col + 4 * lq) = make_float4(fast_atan2_turns(yi[0], yr[0]), fast_atan2_turns(yi[1], yr[1]),
                                                                            fast_atan2_turns(yi[2], yr[2]), fast_atan2_turns(yi[3], yr[3]));
    }
    // the last 64 input samples of the block are the next block's history
    if (tile == tiles - 1 && tid < G::HIST) tail_out[(size_t)c * G::HIST + tid] = load_iq(in_c, (unsigned)(d.N - G::HIST + tid));
}

// =============================================================================================
// k_front_pre_mfma — 1.024 / 2.048 MSa/s, FMD_FLAG_FAST_MATH: k_predecim_mfma and k_front_mfma in one kernel.  fm_in (the phases at
// 256 kSa/s) stays in LDS: the capture is read once and nothing but fm_out and the pilot stage's column sums is written, instead of
// 4 more bytes per 256 kSa/s sample written by one kernel and read by the next (0.27 GB each way per 4096 x 64 ms).
// One workgroup = one station x 1024 fm_out samples (k_front_mfma's tile).  It needs the 2111 phases fm_in[2 o0 - 63 .. 2 o0 + 2047];
// it makes 2112 = 132 columns of 16, one earlier, so that every staged window starts on a multiple of 8 input samples (16-byte loads
// of u8 captures), in NSUB sub-steps of CS columns through one staging array the size of k_predecim_mfma's (the whole input window
// of a tile is 68 KB of cf32 at M = 4).  The loads of sub-step s + 1 are in flight while sub-step s is converted and multiplied.
// The 64 phases in front of a tile are computed twice (3 % more of the first decimator's work; round 2's fused VALU kernel paid 19 %
// on 512-output tiles).  Arithmetic per phase and per fm_out sample is the two kernels' own: outputs are bit-identical to theirs.
// =============================================================================================
template <int M, bool U8>
struct FrontPreGeom {
    using GF = FrontGeomM<1024, 0>;
    static constexpr int NPH = 2112, NCOLS = NPH / 16;            // phases / columns per tile
    static constexpr int NSUB = U8 ? (M == 4 ? 2 : 4) : (M == 4 ? 4 : 6);
    static constexpr int CS = NCOLS / NSUB;                       // columns per sub-step
    static constexpr int SH = 8 - M;
    static constexpr int SEG = 16 * M;                            // input samples between two columns' windows
    static constexpr int NBS = SEG * CS + 64;                     // input samples staged per sub-step
    static constexpr int KS = (64 + SH + 15 * M + 31) / 32;
    static constexpr int NEP = NBS + 8 * (NBS / SEG + 1);         // padded elements per array (see PredecimGeomM)
    static constexpr int NH = U8 ? 1 : 2;
    static constexpr int NWD = NEP / 2;                           // words per array
    static constexpr int ARR_WORDS = (2 * NH * NWD + 3) & ~3;
    static constexpr int LDS_WORDS = ARR_WORDS + GF::NWB;         // staging arrays, then the phases / discriminator output region
    static constexpr int HIST = 64;
    // Measured (4096 stations x 64 ms at 1.024 MSa/s, the kernel on its own / the step): cf32 with the next sub-step's loads in flight over
    // the products and four workgroups per CU 0.45 / 0.64 ms; without 0.47 / 0.67; five per CU (96 registers, spills) 0.47 / 0.72.
    // u8 (not bandwidth-bound: the conversion is the work): five per CU without prefetch 0.26 / 0.45 ms against 0.29 / 0.48 — at M = 4;
    // at M = 8 (six K-steps of operands in registers) that form spills: 0.74 ms a step against 0.61.
    // Two or more tiles per workgroup with the next tile's loads in flight over the back half: 0.56-0.60 on its own, dropped.
    static constexpr bool PREFETCH = !(U8 && M == 4);
    static constexpr int MINWG = (U8 && M == 4) ? 5 : 4;
    static_assert(NCOLS % NSUB == 0 && NBS % 8 == 0 && SEG * (CS - 1) + 32 * KS <= NBS && GF::NWB >= NPH + 1, "geometry");
};
template <int M, typename InT, bool FUSED>
// (launch bounds: the un-fused u8 form — small batches, start-up blocks — takes four workgroups per CU like the others: at five it spilled 24 bytes per lane)
__global__ __launch_bounds__(256, (FUSED ? FrontPreGeom<M, sizeof(InT) == 2>::MINWG : 4)) void k_front_pre_mfma(Dims d, const InT* __restrict__ in, const float2* __restrict__ pre_tail_in, float2* __restrict__ pre_tail_out,
                                                           const float2* __restrict__ tail_in, float2* __restrict__ tail_out, float* __restrict__ fo_pl, float fm_gain,
                                                           const uint4* __restrict__ tab_pre, const uint4* __restrict__ tab, float4* __restrict__ pv_pl,
                                                           const PllSparseTab* __restrict__ sp, PllFusedArgs pf) {
    constexpr bool U8 = sizeof(InT) == 2;
    using G = FrontPreGeom<M, U8>;
    constexpr int NBS = G::NBS, KS = G::KS, SEG = G::SEG, NH = G::NH, NWD = G::NWD, CS = G::CS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = (int)blockIdx.x;
    if constexpr (FUSED) {
        if (bid < pf.n_wg) {
            pll_sparse_body(d, bid * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & (kWave - 1)), pf.pv, pf.hist_in, pf.hist_out, pf.fo, pf.fo_next, pf.poly, pf.poly_next,
                            pf.state, pf.k, pf.tab, 0, pf.spec_stats);
            return;
        }
        bid -= pf.n_wg;
    }
    uint32_t* arr = reinterpret_cast<uint32_t*>(smem);                         // [rail][half][NWD]
    float* region = smem + G::ARR_WORDS;                                       // phases of fm_in[2 o0 - 64 + i]; then the discriminator output
    const int tiles = d.n_fm_out / 1024;
    const int c = bid / tiles, tile = bid % tiles, o0 = tile * 1024, tid = threadIdx.x;
    __builtin_assume(tid >= 0 && tid < 256);
    const int lane = tid & (kWave - 1), wv = tid >> 6, lrow = lane & 15, lq = lane >> 4;
    const int g_first = M * (2 * o0 - 64) + M - 64 - G::SH;                    // first input sample staged (block relative), a multiple of 8
    const InT* in_c = in + (size_t)c * d.N;
    const float2* ptail_c = pre_tail_in + (size_t)c * G::HIST;
    auto word_of = [](int pair) { return pair + 4 * (pair / (SEG / 2)); };      // pair p = elements 2 p, 2 p + 1 -> word of the padded array

    bf16x8 adh[KS], adl[KS];
    auto load_operands = [&]() {
#pragma unroll
        for (int sK = 0; sK < KS; sK++) {
            adh[sK] = __builtin_bit_cast(bf16x8, tab_pre[(sK * 2 + 0) * kWave + lane]);
            adl[sK] = __builtin_bit_cast(bf16x8, tab_pre[(sK * 2 + 1) * kWave + lane]);
        }
    };
    load_operands();
    // items of 16 bytes: two cf32 samples / eight u8 samples.  Samples before the 64 the previous block left (first tile only) feed the
    // phases in front of the block, which come from the phase history below: any in-bounds value does; the last tile's overhang
    // (M = 8: 8 samples) sits under zero taps
    constexpr int IS = U8 ? 8 : 2, ITEMS = NBS / IS, PER = (ITEMS + 255) / 256;
    uint4 buf[PER];
    auto issue1 = [&](int gs, int r) {
        const int j = tid + 256 * r;
        if (j < ITEMS) {
            int g = gs + IS * j;
            g = g < -G::HIST ? -G::HIST : (g < d.N - IS ? g : d.N - IS);
            if constexpr (!U8) buf[r] = (g < 0) ? *reinterpret_cast<const uint4*>(ptail_c + (G::HIST + g)) : ld_stream(reinterpret_cast<const uint4*>(in_c + g));
            else if (g >= 0) buf[r] = ld_stream(reinterpret_cast<const uint4*>(in_c + g));
        }
    };
    auto convert1 = [&](int gs, int r) {
        const int j = tid + 256 * r;
        if (j < ITEMS) {
            if constexpr (!U8) {
                uint32_t h0, l0, h1, l1;
                const int w = word_of(j);
                split_bf16_rn(__uint_as_float(buf[r].x), h0, l0); split_bf16_rn(__uint_as_float(buf[r].z), h1, l1);
                arr[0 * NWD + w] = pack_hi16(h0, h1); arr[1 * NWD + w] = pack_hi16(l0, l1);
                split_bf16_rn(__uint_as_float(buf[r].y), h0, l0); split_bf16_rn(__uint_as_float(buf[r].w), h1, l1);
                arr[2 * NWD + w] = pack_hi16(h0, h1); arr[3 * NWD + w] = pack_hi16(l0, l1);
            } else {
                int g = gs + 8 * j;
                uint32_t re[8], im[8];
                if (g < 0) {        // history, kept as cf32 (u8 captures: integers; after a cf32 block: rounded to bf16)
                    g = g < -G::HIST ? -G::HIST : g;
#pragma unroll
                    for (int u = 0; u < 8; u++) { const float2 v = ptail_c[G::HIST + g + u]; re[u] = f32_bits(v.x) + 0x8000u; im[u] = f32_bits(v.y) + 0x8000u; }
                } else {
                    const uint32_t w4[4] = {buf[r].x, buf[r].y, buf[r].z, buf[r].w};
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t ww = w4[u >> 1] >> (16 * (u & 1));
                        re[u] = f32_bits((float)(ww & 0xffu) - 127.0f); im[u] = f32_bits((float)((ww >> 8) & 0xffu) - 127.0f);      // reference src/app.cpp:56-62
                    }
                }
                const int w = word_of(4 * j);
                *reinterpret_cast<uint4*>(arr + 0 * NWD + w) = make_uint4(pack_hi16(re[0], re[1]), pack_hi16(re[2], re[3]), pack_hi16(re[4], re[5]), pack_hi16(re[6], re[7]));
                *reinterpret_cast<uint4*>(arr + 1 * NWD + w) = make_uint4(pack_hi16(im[0], im[1]), pack_hi16(im[2], im[3]), pack_hi16(im[4], im[5]), pack_hi16(im[6], im[7]));
            }
        }
    };
    if constexpr (G::PREFETCH) {
#pragma unroll
        for (int r = 0; r < PER; r++) issue1(g_first, r);
    }
#pragma unroll 1
    for (int s = 0; s < G::NSUB; s++) {
        if (s) __syncthreads();                                    // the previous sub-step's operand reads
        const int gs = g_first + SEG * CS * s;
        if constexpr (!G::PREFETCH) {
#pragma unroll
            for (int r = 0; r < PER; r++) issue1(gs, r);
        }
#pragma unroll
        for (int r = 0; r < PER; r++) convert1(gs, r);
        if (G::PREFETCH && s + 1 < G::NSUB) {                      // the next sub-step's loads: in flight over this one's products
#pragma unroll
            for (int r = 0; r < PER; r++) issue1(gs + SEG * CS, r);
        }
        __syncthreads();
        for (int ct = wv; ct * 16 < CS; ct += 4) {
            const int col = 16 * ct + lrow, colr = col < CS ? col : CS - 1;
            f32x4 acc[2][3];
#pragma unroll
            for (int sK = 0; sK < KS; sK++) {
                const int e = SEG * colr + 32 * sK + 8 * lq;             // element index, a multiple of 8
                const int w = (e + 8 * (e / SEG)) / 2;
#pragma unroll
                for (int rail = 0; rail < 2; rail++) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arr + (rail * NH) * NWD + w));
                    acc[rail][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adh[sK], bh, sK ? acc[rail][0] : kZero4, 0, 0, 0);
                    acc[rail][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adl[sK], bh, sK ? acc[rail][1] : kZero4, 0, 0, 0);
                    if constexpr (!U8) {
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arr + (rail * NH + 1) * NWD + w));
                        acc[rail][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adh[sK], bl, sK ? acc[rail][2] : kZero4, 0, 0, 0);
                    }
                }
            }
            f32x4 yr, yi;
            if constexpr (U8) { yr = acc[0][0] + acc[0][1]; yi = acc[1][0] + acc[1][1]; }
            else { yr = acc[0][0] + (acc[0][1] + acc[0][2]); yi = acc[1][0] + (acc[1][1] + acc[1][2]); }
            if (col < CS)
                *reinterpret_cast<float4*>(region + 16 * (CS * s + col) + 4 * lq) = make_float4(fast_atan2_turns(yi[0], yr[0]), fast_atan2_turns(yi[1], yr[1]),
                                                                                                fast_atan2_turns(yi[2], yr[2]), fast_atan2_turns(yi[3], yr[3]));
        }
    }
    FrontOps op;
    load_front_ops<0>(op, tab, sp, lane, lq);
    if (tile == 0) {           // the 64 phases in front of the block: the previous block's last ones
        __syncthreads();
        if (tid < 64) region[tid] = tail_in[(size_t)c * d.tail_base + d.tail_base - 64 + tid].x;
    }
    __syncthreads();
    if (tile == tiles - 1) {   // the histories the next block starts from: the last 64 input samples, the last tail_base phases
        if (tid < G::HIST) pre_tail_out[(size_t)c * G::HIST + tid] = load_iq(in_c, (unsigned)(d.N - G::HIST + tid));
        float2* tout = tail_out + (size_t)c * d.tail_base;
        for (int idx = tid; idx < d.tail_base; idx += 256) tout[idx] = make_float2(region[G::NPH - d.tail_base + idx], 0.0f);
    }
    front_from_phases<1024, 0>(d, region, region + 1, reinterpret_cast<uint32_t*>(region), c, o0, tid, fm_gain, nullptr, op, fo_pl, pv_pl, sp);
}

// =============================================================================================
// Lane-per-channel serial kernels.  A wavefront owns 64 adjacent channels; time runs in chunks of
// 32 samples that are loaded row-wise (coalesced, 16 B per lane) and transposed through LDS so each
// lane then walks its own channel.  The next chunk's global loads are in flight while the current
// chunk is processed.
// =============================================================================================
static constexpr int kChunk = 32;
static constexpr int kRowC = kChunk + 2;   // float2 row stride (272 B) of a transposed cf32 chunk
static constexpr int kRowF = kChunk + 4;   // float row stride (144 B) of an f32 output chunk

// 16 x float4 per lane = one 64-channel x 32-sample cf32 chunk in flight per wavefront
struct ChunkRegsC { float4 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15; };

#define FMD_FOR16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

__device__ __forceinline__ ChunkRegsC chunk_load_c(const float2* __restrict__ base, int n, int c0, int C, int t0) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 4, col = lane & 15;
    ChunkRegsC r;
#define FMD_LD(k) { int ch = c0 + 4 * k + row; ch = ch < C ? ch : C - 1; \
                    r.v##k = ld_mid(reinterpret_cast<const float4*>(base + (size_t)ch * n + t0 + 2 * col)); }
    FMD_FOR16(FMD_LD)
#undef FMD_LD
    return r;
}
__device__ __forceinline__ void chunk_store_c(const ChunkRegsC& r, float2* lds) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 4, col = lane & 15;
#define FMD_ST(k) *reinterpret_cast<float4*>(lds + (4 * k + row) * kRowC + 2 * col) = r.v##k;
    FMD_FOR16(FMD_ST)
#undef FMD_ST
}
// flush a transposed cf32 chunk [64][kRowC] to out[C][n] at t0
__device__ __forceinline__ void chunk_flush_c(const float2* lds, float2* __restrict__ out, int n, int c0, int C, int t0) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 4, col = lane & 15;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int r = 4 * k + row, ch = c0 + r;
        if (ch < C) *reinterpret_cast<float4*>(out + (size_t)ch * n + t0 + 2 * col) = *reinterpret_cast<const float4*>(lds + r * kRowC + 2 * col);
    }
}
// flush a [64][kRowF] f32 chunk to out[C][n] at t0
__device__ __forceinline__ void chunk_flush_f(const float* lds, float* __restrict__ out, int n, int c0, int C, int t0) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 3, col = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int r = 8 * k + row, ch = c0 + r;
        if (ch < C) *reinterpret_cast<float4*>(out + (size_t)ch * n + t0 + 4 * col) = *reinterpret_cast<const float4*>(lds + r * kRowF + 4 * col);
    }
}

struct PilotIIR {
    float x1r, x1i, x2r, x2i, y1r, y1i, y2r, y2i;
    // reference IIR_Filter<complex<float>> K=3 (iir_filter.h:40-69) with b=[K,0,0], a=[-r^2, 2r cos, 1]:
    // y = fma(a0, y[n-2], K x[n-2]) + a1 y[n-1]   (the zero-tap terms contribute +-0)
    __device__ __forceinline__ float2 step(float2 x, const LoopCoeffs& k) {
        const float yr = fmaf(k.pilot_a0, y2r, k.pilot_k * x2r) + k.pilot_a1 * y1r;
        const float yi = fmaf(k.pilot_a0, y2i, k.pilot_k * x2i) + k.pilot_a1 * y1i;
        x2r = x1r; x2i = x1i; x1r = x.x; x1i = x.y;
        y2r = y1r; y2i = y1i; y1r = yr; y1i = yi;
        return make_float2(yr, yi);
    }
    __device__ __forceinline__ void load(float* s, int base, int C, int c) {
        x1r = st(s, base + 0, C, c); x1i = st(s, base + 1, C, c); x2r = st(s, base + 2, C, c); x2i = st(s, base + 3, C, c);
        y1r = st(s, base + 4, C, c); y1i = st(s, base + 5, C, c); y2r = st(s, base + 6, C, c); y2i = st(s, base + 7, C, c);
    }
    __device__ __forceinline__ void store(float* s, int base, int C, int c) const {
        st(s, base + 0, C, c) = x1r; st(s, base + 1, C, c) = x1i; st(s, base + 2, C, c) = x2r; st(s, base + 3, C, c) = x2i;
        st(s, base + 4, C, c) = y1r; st(s, base + 5, C, c) = y1i; st(s, base + 6, C, c) = y2r; st(s, base + 7, C, c) = y2i;
    }
};

// a6 + power sum of a7 — reference LockOntoPilot :421-423 (IIR -> pilot_buf) and AGC_Filter::calculate_average_power
// (agc.h:21-30).  Writes the un-gained pilot so the latency-critical PLL kernel does not have to recompute the IIR.
// ROW: the lane's 32-sample row is read from LDS in one go, filtered from registers and written back in one go (no LDS latency
// inside the recurrence: 8 % off the step at 1024 stations, where this kernel's latency is the step) — at the price of ~300
// VGPRs, which costs 2.5 % at 4096 stations (same-box A/B), so larger batches use the sample-by-sample form.
template <bool ROW>
__global__ __launch_bounds__(kWave) void k_pilot_power(Dims d, const float2* __restrict__ fm_out_iq, float2* __restrict__ pilot,
                                                       float* __restrict__ state, LoopCoeffs k, int power_field) {
    __shared__ __attribute__((aligned(16))) float2 xin[kWave * kRowC];
    __builtin_amdgcn_s_setprio(3);  // latency-bound recurrence: win issue arbitration against co-resident FIR waves
    const int lane = threadIdx.x, c0 = blockIdx.x * kWave, c = c0 + lane;
    const bool live = c < d.C;
    const int cs = live ? c : d.C - 1;
    PilotIIR f; f.load(state, SA_X1R, d.C, cs);
    float power = 0.0f;
    const int n = d.n_fm_out, chunks = n / kChunk;     // even: n_fm_out is a multiple of 128
    // TWO chunks in flight in registers (a chunk's 32 IIR steps take ~1.2 us, an HBM load under the pipeline's traffic 2-4 us:
    // with one chunk ahead the kernel sat on the memory latency, 0.52 ms per block alone), one LDS buffer processed in place
    ChunkRegsC ra = chunk_load_c(fm_out_iq, n, c0, d.C, 0);
    ChunkRegsC rb = chunk_load_c(fm_out_iq, n, c0, d.C, kChunk);
#define FMD_POWER_CHUNK(regs, ch_)                                                                                     \
    {                                                                                                                  \
        chunk_store_c(regs, xin);                                                                                      \
        __syncthreads();                                                                                               \
        regs = chunk_load_c(fm_out_iq, n, c0, d.C, ((ch_) + 2 < chunks ? (ch_) + 2 : (ch_)) * kChunk);                  \
        if constexpr (ROW) {                                                                                           \
            float4* row_ = reinterpret_cast<float4*>(xin + lane * kRowC);                                              \
            float4 w_[kChunk / 2];                                                                                     \
            _Pragma("unroll") for (int t = 0; t < kChunk / 2; t++) w_[t] = row_[t];                                    \
            _Pragma("unroll") for (int t = 0; t < kChunk / 2; t++) {                                                   \
                const float2 y0 = f.step(make_float2(w_[t].x, w_[t].y), k);                                            \
                power = power + fmaf(y0.x, y0.x, y0.y * y0.y);                                                         \
                const float2 y1 = f.step(make_float2(w_[t].z, w_[t].w), k);                                            \
                power = power + fmaf(y1.x, y1.x, y1.y * y1.y);                                                         \
                w_[t] = make_float4(y0.x, y0.y, y1.x, y1.y);                                                           \
            }                                                                                                          \
            _Pragma("unroll") for (int t = 0; t < kChunk / 2; t++) row_[t] = w_[t];                                    \
        } else {                                                                                                       \
            _Pragma("unroll 4") for (int t = 0; t < kChunk; t++) {                                                     \
                const float2 y = f.step(xin[lane * kRowC + t], k);                                                     \
                power = power + fmaf(y.x, y.x, y.y * y.y);                                                             \
                xin[lane * kRowC + t] = y;                                                                             \
            }                                                                                                          \
        }                                                                                                              \
        __syncthreads();                                                                                               \
        chunk_flush_c(xin, pilot, n, c0, d.C, (ch_) * kChunk);                                                         \
    }
    for (int ch = 0; ch < chunks; ch += 2) {
        FMD_POWER_CHUNK(ra, ch)
        FMD_POWER_CHUNK(rb, ch + 1)
    }
#undef FMD_POWER_CHUNK
    if (live) { f.store(state, SA_X1R, d.C, c); st(state, power_field, d.C, c) = power; }
}

// a7 gain + a8 — reference LockOntoPilot :423-456, PLL_Mixer::Update (pll_mixer.cpp:12-21)
struct PllState { float lx1, ly1, integ, err, tph; };

// one loop iteration exactly as the reference computes it; returns dt (= the NCO phase in turns)
__device__ __forceinline__ float pll_step(PllState& s, float p, float q, const LoopCoeffs& k) {
    const float Ts = 1.0f / 128000.0f;
    const float KTsI = 0.1f * Ts;
    // loop filter IIR_Filter<float> K=2: t_i = fma(xn[i], b[i], yn[i]*a[i]); y += t_i
    const float t0 = fmaf(s.lx1, k.pll_b0, s.ly1 * k.pll_a0);
    const float t1 = fmaf(s.err, k.pll_b1, 0.0f);
    const float lpf = (0.0f + t0) + t1;
    s.lx1 = s.err; s.ly1 = lpf;
    const float P = lpf * 0.01f;
    s.integ = clampf(fmaf(s.err, KTsI, s.integ), -1.0f, 1.0f);
    const float PI_error = s.integ + P;
    // PLL_Mixer::Update
    const float control = clampf(PI_error * 1.0f, -1.0f, 1.0f);
    const float freq = fmaf(control, -100.0f, -19000.0f);
    const float yy = fmaf(freq, Ts, s.tph);
    s.tph = yy - round_half_away(yy);
    float dt_cos = s.tph + 0.25f;
    dt_cos = dt_cos - round_half_away(dt_cos);
    const float ps = cheb_sine_scalar(s.tph);
    const float pc = cheb_sine_scalar(dt_cos);
    const float res_im = fmaf(ps, p, q * pc);
    const float res_re = fmaf(p, pc, -(q * ps));
    s.err = fmd_atan2f(res_im, res_re);
    return s.tph;
}

// FMD_FLAG_KEEP_TAPS, exact mode: the loop's per-sample traces behind the reference's GetPilotOutput (the pilot after its AGC),
// GetPLLOutput (the NCO's (cos, sin)), Get_PLL_Raw_Phase_Error_Output and Get_PLL_LPF_Phase_Error_Output (broadcast_fm_demod.h:245-248;
// what the oracle keeps at fm_oracle.c lock_onto_pilot).  The time-parallel kernels below commit only the NCO phase; this is the plain
// reference iteration once more, lane per station, ahead of them on the same stream from the same start state (it writes no state).
// A getter's kernel: uncoalesced by design, launched only for handles that asked for the traces.
__global__ __launch_bounds__(kWave) void k_pll_taps(Dims d, const float2* __restrict__ pilot, const float* __restrict__ state, LoopCoeffs k, int power_field, TapPtrs t) {
    const int c = blockIdx.x * kWave + threadIdx.x;
    if (c >= d.C) return;
    const int n = d.n_fm_out;
    float gain = state[(size_t)S_AGC_PILOT_GAIN * d.C + c];
    {   // AGC_Filter::process (agc.h:12-19) as k_pilot_pll computes it
        const float sum = state[(size_t)power_field * d.C + c];
        const float target_gain = sqrtf((1.0f / sum) * (float)n);
        gain = fmaf(target_gain - gain, 0.2f, gain);
    }
    PllState S{state[(size_t)S_PLL_X1 * d.C + c], state[(size_t)S_PLL_Y1 * d.C + c], state[(size_t)S_PLL_INT * d.C + c], state[(size_t)S_PLL_ERR * d.C + c],
               state[(size_t)S_PLL_T * d.C + c]};
    const float Ts = 1.0f / 128000.0f, KTsI = 0.1f * Ts;
    for (int i = 0; i < n; i++) {
        const float2 x = pilot[(size_t)c * n + i];
        const float p = gain * x.x, q = gain * x.y;
        // (pll_step, with what it computes on the way kept)
        const float t0 = fmaf(S.lx1, k.pll_b0, S.ly1 * k.pll_a0);
        const float t1 = fmaf(S.err, k.pll_b1, 0.0f);
        const float lpf = (0.0f + t0) + t1;
        S.lx1 = S.err; S.ly1 = lpf;
        const float P = lpf * 0.01f;
        S.integ = clampf(fmaf(S.err, KTsI, S.integ), -1.0f, 1.0f);
        const float PI_error = S.integ + P;
        const float control = clampf(PI_error * 1.0f, -1.0f, 1.0f);
        const float freq = fmaf(control, -100.0f, -19000.0f);
        const float yy = fmaf(freq, Ts, S.tph);
        S.tph = yy - round_half_away(yy);
        float dt_cos = S.tph + 0.25f;
        dt_cos = dt_cos - round_half_away(dt_cos);
        const float ps = cheb_sine_scalar(S.tph), pc = cheb_sine_scalar(dt_cos);
        S.err = fmd_atan2f(fmaf(ps, p, q * pc), fmaf(p, pc, -(q * ps)));
        const size_t o = (size_t)c * n + i;
        t.pilot[o] = make_float2(p, q); t.pll[o] = make_float2(pc, ps); t.pll_raw[o] = S.err; t.pll_pi[o] = PI_error;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_pilot_pll — frequency-speculative, time-parallel evaluation of the pilot PLL.
//
// The loop is a strictly serial recurrence: err_{t-1} -> loop filter -> frequency word F_t -> NCO phase tph_t -> phase
// detector -> err_t, 78 dependent operations per sample as the reference writes it, and a lone wavefront spends ~6 cycles on
// every dependent operation (4 on every instruction) whatever its width.  But F_t is a float near -19000 (ulp 2^-9) driven by
// a heavily low-passed error: in lock it changes on ~0.5 % of samples.  So sixteen lanes evaluate sixteen consecutive samples
// of one channel AT ONCE under the assumption "F stays what it is for the first of them", and the assumption is then checked
// exactly:
//   (A) every lane: S_0 = U(state, err_prev), F = frequency word of the span's first sample  (exact, no assumption)
//   (B) every lane: the phase recurrence tph <- wrap(tph + F Ts), 16 steps; lane j keeps step j     (3 dependent ops per step)
//   (C) lane j: its own sample's phase detector -> err_j          (the expensive part: two chebyshev sines, atan2; in parallel)
//   (D) every lane: the loop filter over err_0..err_14 (4 dependent ops per step); lane i keeps the filter state S_i and
//       decides whether S_i still yields F.  A ballot gives the group the index m of its first sample that does not:
//       samples 0..m-1 are exactly what the serial loop would have produced (sample 0 always is) and are committed; the
//       state to resume from sits in lane m-1 and is fetched with ds_bpermute.  The next span starts at sample m.
// Nothing is approximated and nothing is replayed on a mis-speculation: a span just commits fewer samples (15.2 of 16 on
// average in lock).  The serial work left per sample is (B) + (D), 7 dependent operations instead of 78.
//
// (C) uses short forms that equal the reference arithmetic on a locked loop's operands — x - rndne(x) for the phase wraps
// (equal unless x is an exact tie, visible as z - 1/4 == 0 in the chebyshev argument), atan2f's first range with an unscaled
// division (x in [2^-28, ~2e8), (y/x)^2 in [2^-58, 49/256): both windows 0x1bc40000 wide in the bit pattern) — and every lane
// reports whether its operands were inside; if a lane that matters was not, the whole wavefront redoes (B)+(C) for that span
// with the reference forms.  (D) runs the integrator unclamped and checks at both ends of the span that the clamp could not
// have acted (it moves < 4e-6 per sample); otherwise the span is verified with the exact predicated loop.
//
// A loop out of lock (acquisition, a station without pilot) commits ~1 sample per span under that assumption.  Round 3: after a
// chunk that needed more than 3x the fewest spans the wavefront runs the next chunks with the plain serial iteration (pll_step),
// backing off exponentially — 3.7-3.9 ms a block for a batch with ANY such station against 0.76 in lock.  Round 6: the word still
// moves by only a few ulp a sample, so such a wavefront speculates on the SEQUENCE of words instead (a cheap guess pass, then the
// exact pass confirms word by word: pilot_pll_body's span, seq): K samples a span at ~1.4x a span's cost, 1.06-1.25 ms a block (4096 stations,
// 16 lanes while wavefronts are out of lock: fmd_api.cpp picks the lane count, and this kernel over the low-work one, by what the kernels report).
// The kernel holds two chunk loops — round 3's unchanged for wavefronts whose stations held lock through the previous block, the
// sequence-capable one for the others (Buffers::pll_hint, per station, written by the wavefront itself) — so the all-locked batch pays next to nothing.
//
// Layout: one wavefront = 64 / K channels x K lanes (K = 16 or 8), one workgroup = one wavefront (12.6 KB LDS and 128 VGPRs
// at K = 16: fits any hole a retiring FIR workgroup leaves; 24.8 KB and 169 VGPRs at K = 8).  128-sample chunks; LDS rings of two chunks per channel for the pilot samples and the
// results; the chunk after next is in flight in 4 float4 registers per lane; all global traffic is 16-byte, row-contiguous.
// Constants live in VGPRs (a 32-bit literal costs a lone wave ~2.7 cycles per instruction).
// ---------------------------------------------------------------------------------------------------------------
struct PllConsts {
    float b0, a0, b1, c001, ktsi, m100, m19000, ts, q25, mq25, c5, c4, c3, c2, c1, c0;
    float a10, a8, a6, a4, a2, a0t, a9, a7, a5, a3, a1;
    uint32_t xlo, zlo;
};
#define FMD_OPAQUE_F(dst, val) { float t_ = (val); asm volatile("" : "+v"(t_)); dst = t_; }
#define FMD_OPAQUE_U(dst, val) { uint32_t t_ = (val); asm volatile("" : "+v"(t_)); dst = t_; }
__device__ __forceinline__ PllConsts make_pll_consts(const LoopCoeffs& k) {
    PllConsts c;
    FMD_OPAQUE_F(c.b0, k.pll_b0) FMD_OPAQUE_F(c.a0, k.pll_a0) FMD_OPAQUE_F(c.b1, k.pll_b1) FMD_OPAQUE_F(c.c001, 0.01f)
    FMD_OPAQUE_F(c.ktsi, 0.1f * (1.0f / 128000.0f)) FMD_OPAQUE_F(c.m100, -100.0f) FMD_OPAQUE_F(c.m19000, -19000.0f)
    FMD_OPAQUE_F(c.ts, 1.0f / 128000.0f) FMD_OPAQUE_F(c.q25, 0.25f) FMD_OPAQUE_F(c.mq25, -0.25f)
    FMD_OPAQUE_F(c.c5, 3.20396066f) FMD_OPAQUE_F(c.c4, -14.07150173f) FMD_OPAQUE_F(c.c3, 38.50016403f)
    FMD_OPAQUE_F(c.c2, -67.07687378f) FMD_OPAQUE_F(c.c1, 64.83583069f) FMD_OPAQUE_F(c.c0, -25.13274193f)
    FMD_OPAQUE_F(c.a10, bits_f32(0x3c8569d7u)) FMD_OPAQUE_F(c.a8, bits_f32(0x3d4bda59u)) FMD_OPAQUE_F(c.a6, bits_f32(0x3d886b35u))
    FMD_OPAQUE_F(c.a4, bits_f32(0x3dba2e6eu)) FMD_OPAQUE_F(c.a2, bits_f32(0x3e124925u)) FMD_OPAQUE_F(c.a0t, bits_f32(0x3eaaaaabu))
    FMD_OPAQUE_F(c.a9, bits_f32(0xbd15a221u)) FMD_OPAQUE_F(c.a7, bits_f32(0xbd6ef16bu)) FMD_OPAQUE_F(c.a5, bits_f32(0xbd9d8795u))
    FMD_OPAQUE_F(c.a3, bits_f32(0xbde38e38u)) FMD_OPAQUE_F(c.a1, bits_f32(0xbe4ccccdu))
    FMD_OPAQUE_U(c.xlo, 0x31800000u) FMD_OPAQUE_U(c.zlo, 0x22800000u)
    return c;
}
static constexpr uint32_t kRangeWindow = 0x1bc40000u;   // bits(49/256) - bits(2^-58)

// chebyshev_sine (scalar association) with register constants; zq = z - 1/4 is returned for the tie test
__device__ __forceinline__ float cheb_sine_locked(float x, const PllConsts& c, float& zq) {
    const float z = x * x;
    float p = fmaf(c.c5, z, c.c4);
    p = fmaf(p, z, c.c3); p = fmaf(p, z, c.c2); p = fmaf(p, z, c.c1); p = fmaf(p, z, c.c0);
    zq = z + c.mq25;
    return (zq * x) * p;
}

// atan2f(y, x) for x in [2^-28, ~2e8) and 2^-29 <= |y/x| < 7/16 (a locked loop's phase error): the published algorithm
// reduces to t - t (s1 + s2) with t = y / x (its first range, odd-symmetric: no quadrant or sign selects), and the division
// needs neither operand scaling nor special-value fix-up.  ok reports whether the operands were inside.
__device__ __forceinline__ float atan2f_locked(float y, float x, const PllConsts& c, bool& ok) {
    const float t = div_unscaled(y, x);
    const float z = t * t, w = z * z;
    ok = max(f32_bits(x) - c.xlo, f32_bits(z) - c.zlo) < kRangeWindow;
    float s1 = c.a8 + w * c.a10; s1 = c.a6 + w * s1; s1 = c.a4 + w * s1; s1 = c.a2 + w * s1; s1 = c.a0t + w * s1; s1 = z * s1;
    float s2 = c.a7 + w * c.a9; s2 = c.a5 + w * s2; s2 = c.a3 + w * s2; s2 = c.a1 + w * s2; s2 = w * s2;
    return t - t * (s1 + s2);
}

template <int S>
__device__ __forceinline__ float dpp_row_shr(float v) {      // lane l of a 16-lane row gets lane l - S's value (0 below the row's first lane)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + S, 0xf, 0xf, true));
}

static constexpr int kPllChunk = 128;         // samples per chunk (64 for the 8-lane variant, to halve its 24.5 KB of LDS, cost 10 % of the step)
static constexpr int kPllRing = 2 * kPllChunk;
static constexpr int kSlowHoldMax = 16;       // longest run of serial chunks between two speculation attempts

// K = lanes (= consecutive samples) per channel.  K = 16 gives the shortest latency (4 channels per wavefront, 1024 wavefronts
// for 4096 channels); K = 8 spends 30 % fewer VALU instructions (8 channels per wavefront share the serial parts) for ~25 %
// more latency — better as soon as the chip, not a lone wavefront, is the limit.
template <int K>
__device__ __forceinline__ void pilot_pll_body(Dims d, const float2* __restrict__ pilot, float* __restrict__ pll_dt,
                                               float* __restrict__ state, LoopCoeffs k, int power_field,
                                               unsigned long long* __restrict__ spec_stats,
                                               unsigned int* __restrict__ chain, unsigned int seq, unsigned int* __restrict__ hint, unsigned int launch_no,
                                               float2 (*xin)[kPllRing], float (*dts)[kPllRing], float (*ex)[K + 4], float (*e1x)[K + 4], float (*fsq)[K + 4]) {
    constexpr int G = kWave / K, CH = kPllChunk, RING = kPllRing;
    constexpr int kPllSlowSpans = 3 * CH / K;     // a chunk that needed more spans than this is "out of lock"
    constexpr int kPllStuckSpans = 16;            // sequence-form spans in a row that committed one sample: serial chunks next
    constexpr int kPllSeqSpans = (18 * CH) / (10 * K);
    __builtin_amdgcn_s_setprio(3);  // latency-bound recurrence: win issue arbitration against co-resident FIR waves
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x, g = lane / K, j = lane % K;
    const int c0 = blockIdx.x * G, c = c0 + g;
    const bool live = c < d.C;
    const int cs = live ? c : d.C - 1;
    const int n = d.n_fm_out, chunks = n / CH;
    const PllConsts kc = make_pll_consts(k);

    // global <-> LDS: lane (g, j) moves row g; chunk q of the pilot = 4 float4 (2 samples each) per lane
    const int srow = c0 + g < d.C ? c0 + g : d.C - 1;
    const float2* prow = pilot + (size_t)srow * n;
    constexpr int NLD = CH / (2 * K), NST = CH / (4 * K);   // float4 loads / stores per lane and chunk
    // named registers, not an array: an indexed local array ends up in scratch memory here
    float4 pre0{}, pre1{}, pre2{}, pre3{}, pre4{}, pre5{}, pre6{}, pre7{};
#define FMD_PLL_EACH(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define FMD_PLL_LD(r) if constexpr (r < NLD) pre##r = *reinterpret_cast<const float4*>(prow + b0_ + 2 * K * r);
#define FMD_PLL_ST(r) if constexpr (r < NLD) *reinterpret_cast<float4*>(row_ + 2 * K * r) = pre##r;
#define FMD_PLL_FETCH(q_) { const int b0_ = ((q_) < chunks ? (q_) : chunks - 1) * CH + 2 * j; FMD_PLL_EACH(FMD_PLL_LD) }
#define FMD_PLL_STASH(q_) { float2* row_ = &xin[g][((q_) & 1) * CH + 2 * j]; FMD_PLL_EACH(FMD_PLL_ST) }
    FMD_PLL_FETCH(0) FMD_PLL_STASH(0)
    FMD_PLL_FETCH(1) FMD_PLL_STASH(1)
    FMD_PLL_FETCH(2)

    // Block-to-block hand-over per WAVEFRONT instead of per kernel.  The loop state of these channels is written by the same
    // wavefront of the previous block's launch; with `chain` the launches of consecutive blocks sit on two streams, this one
    // may start (and has its first chunks in flight, above) while the previous one still runs, and waits here until its
    // predecessor wavefront has published its state: chain[wave] == seq - 1 (release/acquire at agent scope).  No queue
    // packet, no kernel boundary and no launch ramp between the two ends of the serial chain any more.  The predecessor was
    // submitted earlier and never waits on anything later, and 2 x 512 such workgroups cannot fill the chip, so the wait
    // cannot deadlock; a 2 s watchdog turns a lost predecessor (a failed launch) into an error flag instead of a hang.
    // Used for effective batches up to 3328 stations (fmd_api.cpp, effective_channels): beyond, two launches' workgroups resident at once take more
    // from the FIR kernels than the hand-over gap gives back (same-box A/B: -13 % at 1024 stations, -7 % at 3072, +4 % at 4096).
    if (chain) {
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        // relaxed polls (an acquire load invalidates the caches at agent scope every time round: with hundreds of waiting
        // wavefronts that slows every kernel on the chip), one acquire fence once the predecessor has published
        while (__hip_atomic_load(&chain[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq - 1u) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - w0 > 200000000ull) {   // 100 MHz ticks
                if (lane == 0) __hip_atomic_store(&chain[gridDim.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // AGC_Filter::process (agc.h:12-19) as compiled: target_gain = sqrt((target/sum) * N)
    float gain = st(state, S_AGC_PILOT_GAIN, d.C, cs);
    {
        const float sum = st(state, power_field, d.C, cs);
        const float target_gain = sqrtf((1.0f / sum) * (float)n);
        gain = fmaf(target_gain - gain, 0.2f, gain);
    }
    // loop state, identical in the 16 lanes of a channel
    float lx1 = st(state, S_PLL_X1, d.C, cs), ly1 = st(state, S_PLL_Y1, d.C, cs), integ = st(state, S_PLL_INT, d.C, cs);
    float err_prev = st(state, S_PLL_ERR, d.C, cs), tph_prev = st(state, S_PLL_T, d.C, cs);
    ex[g][0] = err_prev;
    // which of the two loops below these stations get: round 3's while they held lock through the previous block (one flag per station, written at the end)
    const bool SEQCAP = hint && __builtin_amdgcn_ballot_w64(live && hint[cs] != 0u) != 0ull;

    int pos = 0;                 // next sample of this channel (absolute within the block); identical in its 16 lanes
    int seq_left = 0, hold = 0;  // wave-uniform: serial chunks still to run / back-off
    unsigned long long n_exact = 0, n_seq = 0;   // wave-uniform counters (scalar registers)
    int n_spans = 0;
    bool seq_form = false;       // (second loop) wave-uniform: the spans run in the sequence form (below)
    unsigned long long n_seq_spans = 0;
    int q_resume = 0, slow_run = 0;   // the first chunk for the second loop; slow chunks in a row
    bool handed_over = false;    // the first loop met a loop out of lock and left the rest of the block to the second
    if (!SEQCAP) {
    // ---- the constant-word form (round 3's span), for a wavefront whose stations all held lock through the previous block.  Two chunks in a row that
    // needed more than 3x the fewest spans say a loop is out of lock: the rest of the block goes to the second loop (round 3 ran serial chunks from
    // the first such chunk on: 3.6 ms for the block in which a station loses lock; now ~1.3) ----
    for (int q = 0; q < chunks; q++) {
        const int cend = (q + 1) * CH;
        int spans = 0;
        {
            while (__builtin_amdgcn_ballot_w64(pos < cend) != 0ull) {
                const bool active = pos < cend;
                const int rem = n - pos;                             // samples left in the block for this channel
                spans++;
                // (A) S_0 = U(state, err_prev) and the exact frequency word of the span's first sample
                float y1, ig;
                {
                    const float t0 = fmaf(lx1, kc.b0, ly1 * kc.a0);
                    const float t1 = fmaf(err_prev, kc.b1, 0.0f);
                    y1 = (0.0f + t0) + t1;
                    ig = clampf(fmaf(err_prev, kc.ktsi, integ), -1.0f, 1.0f);
                }
                const float F = fmaf(clampf(ig + y1 * kc.c001, -1.0f, 1.0f), kc.m100, kc.m19000);
                // (B) phase recurrence with constant F; lane j keeps the phase after j+1 steps
                float tph = tph_prev, mine = 0.0f;
#pragma unroll
                for (int i = 0; i < K; i++) {
                    const float yy = fmaf(F, kc.ts, tph);
                    tph = yy - rintf(yy);
                    mine = (i == j) ? tph : mine;
                }
                // (C) this lane's sample: phase detector with the locked short forms
                const int t = active ? pos + j : j;
                const float2 x = xin[g][t & (RING - 1)];
                const float p = gain * x.x, q2 = gain * x.y;
                const float dc = mine + kc.q25;
                const float dt_cos = dc - rintf(dc);
                float zq_s, zq_c;
                const float ps = cheb_sine_locked(mine, kc, zq_s), pc = cheb_sine_locked(dt_cos, kc, zq_c);
                bool in_range;
                float e = atan2f_locked(fmaf(ps, p, q2 * pc), fmaf(p, pc, -(q2 * ps)), kc, in_range);
                const bool lane_ok = in_range && (fminf(fabsf(zq_s), fabsf(zq_c)) != 0.0f);
                if (__builtin_amdgcn_ballot_w64(active && !lane_ok && j < rem) != 0ull) {
                    // a short form was outside its domain somewhere (loop out of lock, or an exact tie): reference forms for all
                    n_exact++;
                    float tp = tph_prev, mm = 0.0f;
                    for (int i = 0; i < K; i++) { const float yy = fmaf(F, kc.ts, tp); tp = yy - round_half_away(yy); mm = (i == j) ? tp : mm; }
                    float dcg = mm + 0.25f; dcg = dcg - round_half_away(dcg);
                    const float psg = cheb_sine_scalar(mm), pcg = cheb_sine_scalar(dcg);
                    e = fmd_atan2f(fmaf(psg, p, q2 * pcg), fmaf(p, pcg, -(q2 * psg)));
                    mine = mm;
                }
                ex[g][j + 1] = e;
                e1x[g][j] = fmaf(e, kc.b1, 0.0f);
                if (active) dts[g][t & (RING - 1)] = mine;           // speculative; samples past the commit point are rewritten
                float ev[K], t1v[K];
#pragma unroll
                for (int i = 0; i < K; i++) { ev[i] = ex[g][i + 1]; t1v[i] = e1x[g][i]; }
                // (D) loop filter over the span, identically in every lane of the channel; lane i keeps S_i = (y1_i, ig_i)
                float fy1 = y1, fig = ig, fx1 = err_prev, my_y1 = y1, my_ig = ig;
#pragma unroll
                for (int i = 1; i < K; i++) {
                    const float t0 = fmaf(fx1, kc.b0, fy1 * kc.a0);
                    fy1 = (0.0f + t0) + t1v[i - 1]; fx1 = ev[i - 1];
                    fig = fmaf(ev[i - 1], kc.ktsi, fig);
                    my_y1 = (i == j) ? fy1 : my_y1; my_ig = (i == j) ? fig : my_ig;
                }
                // the integrator moves < 4e-6 per sample: its clamp acted nowhere in the span iff both ends are well inside
                const bool integ_clamped = !((fabsf(ig) <= 0.99f) && (fabsf(fig) <= 0.99f));
                const float Fj = fmaf(clampf(my_ig + my_y1 * kc.c001, -1.0f, 1.0f), kc.m100, kc.m19000);
                const bool ok_j = (f32_bits(Fj) == f32_bits(F)) && (j < rem);
                const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok_j || j == 0);
                int m = __builtin_ctz(~((unsigned int)(okm >> (g * K)) & ((1u << K) - 1u)) | (1u << K));   // first invalid sample (K = none)
                float nx, ny, ni, ne, nt;
                if (__builtin_amdgcn_ballot_w64(active && integ_clamped) != 0ull) {
                    // a saturated integrator (never in lock): verify with the exact clamps, predicated
                    float x1 = err_prev, yy1 = y1, ig2 = ig;
                    bool valid = true;
                    m = 1;
                    for (int i = 1; i < K; i++) {
                        const float ei = ex[g][i];
                        const float t0 = fmaf(x1, k.pll_b0, yy1 * k.pll_a0), t1 = fmaf(ei, k.pll_b1, 0.0f);
                        const float ny1 = (0.0f + t0) + t1;
                        const float ni1 = clampf(fmaf(ei, 0.1f * (1.0f / 128000.0f), ig2), -1.0f, 1.0f);
                        const float Fi = fmaf(clampf((ni1 + ny1 * 0.01f) * 1.0f, -1.0f, 1.0f), -100.0f, -19000.0f);
                        valid = valid && (f32_bits(Fi) == f32_bits(F)) && (i < rem);
                        if (valid) { x1 = ei; yy1 = ny1; ig2 = ni1; m = i + 1; }
                    }
                    nx = x1; ny = yy1; ni = ig2; ne = ex[g][m]; nt = dts[g][(pos + m - 1) & (RING - 1)];
                } else {
                    // resume state S_{m-1}, err_{m-1}, tph_{m-1}: held by lane m-1 of the channel
                    const int src = (g * K + m - 1) * 4;
                    ny = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(my_y1)));
                    ni = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(my_ig)));
                    ne = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(e)));
                    nt = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(mine)));
                    nx = ex[g][m - 1];                               // err_{m-2} (ex[g][0] = err_prev)
                }
                if (active) {
                    lx1 = nx; ly1 = ny; integ = ni; err_prev = ne; tph_prev = nt; pos += m;
                }
                ex[g][0] = err_prev;
            }
        }
        // chunk q is final in every channel: drain it, refill its half of the ring with chunk q+2, prefetch chunk q+3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {   // results of chunk q
            const float* row = &dts[g][(q & 1) * CH + 4 * j];
            float* o = pll_dt + (size_t)cs * n + (size_t)q * CH + 4 * j;
#define FMD_PLL_DRAIN(r) if constexpr (r < NST) { const float4 v_ = *reinterpret_cast<const float4*>(row + 4 * K * r); if (live) *reinterpret_cast<float4*>(o + 4 * K * r) = v_; }
            FMD_PLL_EACH(FMD_PLL_DRAIN)
        }
        FMD_PLL_STASH(q + 2)
        FMD_PLL_FETCH(q + 3)
        n_spans += spans;
        slow_run = spans > kPllSlowSpans ? slow_run + 1 : 0;        // (two in a row: a single slow chunk is what a loop in lock has now and then)
        if (slow_run >= 2) { q_resume = q + 1; handed_over = true; break; }
    }
    }
    if (SEQCAP || handed_over) {
    // ---- the same with the sequence form for loops out of lock (round 6): a wavefront that left the previous block out of lock, or met such a loop above ----
    seq_form = handed_over;
    int stuck = 0;               // wave-uniform: consecutive sequence-form spans in which some channel committed one sample
    // One span of every channel of the wavefront: (A)-(D) above.
    // seq (round 6) — the form for a loop OUT of lock (a station without a pilot, acquisition): the frequency word then moves on every sample
    // and a span under "F stays put" commits one sample, which is why such a wavefront used to fall back to the 78-operation serial
    // iteration for whole chunks (4 ms a block: the exact mode's cliff).  But the word moves by only a few ulp a sample —
    // it is the heavily low-passed error scaled by 1e-2 x 1e2 next to -19000, one ulp = 2^-9 Hz is a radian of phase error — and a
    // phase that is wrong by those few ulp x Ts changes the next errors by 1e-6 rad, the word by 1e-6 of an ulp.  So the span is
    // evaluated twice: a GUESS pass (constant F, hardware sine / cosine, minimax arctangent, the loop filter) whose only product is the
    // sequence of words F_0 .. F_{K-1}, and the EXACT pass — (B) stepping the phase with that sequence, (C) in the reference's forms,
    // (D) confirming word by word that the filter state behind sample i-1 yields exactly the F_i the phase was stepped with.  The
    // induction is the same as for a constant word (sample 0's word is exact by construction), so is the commit; a wrong guess costs
    // samples, never bits.  ~2.3 spans' worth of instructions per K samples instead of K serial iterations.
    // One body with a wave-uniform flag, not two instantiations: the exact pass IS the reference-forms branch the other form falls back to
    // (with the kernel's code doubled, the rare excursions into the second copy cost an all-locked batch 12 %: instruction cache).
    // Returns (wave-uniform, seq only): few — some channel committed one sample; changes — how often the guessed words changed inside the span on
    // the channel where they changed most (up to 4): "F stays put" would have needed about 1 + changes spans for these samples.
    float sc_a[4], sc_1[4], sc_pw = 1.0f;                  // the scans' coefficients for this lane: shift s = 1, 2, 4, 8 reaches lane j - s of the same channel or nothing
    {
        float as = kc.a0;
#pragma unroll
        for (int t = 0; t < 4; t++) { sc_a[t] = j >= (1 << t) ? as : 0.0f; sc_1[t] = j >= (1 << t) ? 1.0f : 0.0f; as = as * as; }
        for (int t = 0; t < j; t++) sc_pw *= kc.a0;       // a0^j
    }
    auto span = [&](const bool seq, const int cend, bool& few, int& changes) __attribute__((always_inline)) {
        const bool active = pos < cend;
        const int rem = n - pos;                             // samples left in the block for this channel
        // (A) S_0 = U(state, err_prev) and the exact frequency word of the span's first sample
        float y1, ig;
        {
            const float t0 = fmaf(lx1, kc.b0, ly1 * kc.a0);
            const float t1 = fmaf(err_prev, kc.b1, 0.0f);
            y1 = (0.0f + t0) + t1;
            ig = clampf(fmaf(err_prev, kc.ktsi, integ), -1.0f, 1.0f);
        }
        const float F = fmaf(clampf(ig + y1 * kc.c001, -1.0f, 1.0f), kc.m100, kc.m19000);
        const int t = active ? pos + j : j;
        const float2 x = xin[g][t & (RING - 1)];
        const float p = gain * x.x, q2 = gain * x.y;
        float Fq[K];                                         // seq: the word step i of (B) uses
        float Fmine = F;                                     // ... and the one this lane's sample was stepped into with: what (D) confirms
        float mine = 0.0f, e = 0.0f;
        if (seq) {
            // the guess pass
            const float mg = fmaf((float)(j + 1), F * kc.ts, tph_prev);      // (the phase under a constant word, unwrapped: the hardware sine takes turns)
            const float psg = fast_sin_turns(mg), pcg = fast_cos_turns(mg);
            const float eg = fast_atan2f(fmaf(psg, p, q2 * pcg), fmaf(p, pcg, -(q2 * psg)));
            // the loop filter over the guessed errors, in closed form on the lanes (a guess: any summation order will do) instead of the serial
            // recurrence through LDS: lane l holds u_l = b1 e_l + b0 e_(l-1); the states behind sample j - 1 are
            //   y1_j = a0^j y1_0 + sum_(k < j) a0^k u_(j-1-k),   ig_j = ig_0 + K Ts sum_(l < j) e_l
            // — two inclusive scans by doubling over the channel's lanes (row shifts by 1, 2, 4, 8; the coefficients are zero where a shift
            // would reach below the channel's first lane), then one more shift by a lane.
            const float e_below = dpp_row_shr<1>(eg);
            float vy = fmaf(kc.b1, eg, kc.b0 * (j == 0 ? err_prev : e_below)), vi = eg;
            vy = fmaf(sc_a[0], dpp_row_shr<1>(vy), vy); vi = fmaf(sc_1[0], dpp_row_shr<1>(vi), vi);
            vy = fmaf(sc_a[1], dpp_row_shr<2>(vy), vy); vi = fmaf(sc_1[1], dpp_row_shr<2>(vi), vi);
            vy = fmaf(sc_a[2], dpp_row_shr<4>(vy), vy); vi = fmaf(sc_1[2], dpp_row_shr<4>(vi), vi);
            if constexpr (K == 16) { vy = fmaf(sc_a[3], dpp_row_shr<8>(vy), vy); vi = fmaf(sc_1[3], dpp_row_shr<8>(vi), vi); }
            const float vy_below = dpp_row_shr<1>(vy), vi_below = dpp_row_shr<1>(vi);
            const float my_gy1 = fmaf(sc_pw, y1, j == 0 ? 0.0f : vy_below);
            const float my_gig = clampf(fmaf(kc.ktsi, j == 0 ? 0.0f : vi_below, ig), -1.0f, 1.0f);
            Fmine = fmaf(clampf(my_gig + my_gy1 * kc.c001, -1.0f, 1.0f), kc.m100, kc.m19000);    // (lane 0: S_0's word, F itself)
            fsq[g][j] = Fmine;
#pragma unroll
            for (int i = 0; i < K; i++) Fq[i] = fsq[g][i];
        }
        bool reference_forms;
        {
            // (B) phase recurrence with the word(s); lane j keeps the phase after j+1 steps
            float tph = tph_prev;
#pragma unroll
            for (int i = 0; i < K; i++) {
                const float yy = fmaf(seq ? Fq[i] : F, kc.ts, tph);
                tph = yy - rintf(yy);
                mine = (i == j) ? tph : mine;
            }
            // (C) this lane's sample: phase detector with the short wraps and sines; the arctangent's short form for a loop in lock, the
            // reference's in the sequence form (its operands are anywhere)
            const float dc = mine + kc.q25;
            const float dt_cos = dc - rintf(dc);
            float zq_s, zq_c;
            const float ps = cheb_sine_locked(mine, kc, zq_s), pc = cheb_sine_locked(dt_cos, kc, zq_c);
            bool in_range = true;
            const float res_im = fmaf(ps, p, q2 * pc), res_re = fmaf(p, pc, -(q2 * ps));
            if (seq) e = fmd_atan2f(res_im, res_re);
            else e = atan2f_locked(res_im, res_re, kc, in_range);
            const bool lane_ok = in_range && (fminf(fabsf(zq_s), fabsf(zq_c)) != 0.0f);
            // a short form was outside its domain somewhere (loop out of lock, or an exact tie): reference forms for all
            reference_forms = __builtin_amdgcn_ballot_w64(active && !lane_ok && j < rem) != 0ull;
        }
        if (reference_forms) {
            n_exact++;
            float tp = tph_prev, mm = 0.0f;
#pragma unroll
            for (int i = 0; i < K; i++) { const float yy = fmaf(seq ? Fq[i] : F, kc.ts, tp); tp = yy - round_half_away(yy); mm = (i == j) ? tp : mm; }
            float dcg = mm + 0.25f; dcg = dcg - round_half_away(dcg);
            const float psg = cheb_sine_scalar(mm), pcg = cheb_sine_scalar(dcg);
            e = fmd_atan2f(fmaf(psg, p, q2 * pcg), fmaf(p, pcg, -(q2 * psg)));
            mine = mm;
        }
        ex[g][j + 1] = e;
        e1x[g][j] = fmaf(e, kc.b1, 0.0f);
        if (active) dts[g][t & (RING - 1)] = mine;           // speculative; samples past the commit point are rewritten
        float ev[K], t1v[K];
#pragma unroll
        for (int i = 0; i < K; i++) { ev[i] = ex[g][i + 1]; t1v[i] = e1x[g][i]; }
        // (D) loop filter over the span, identically in every lane of the channel; lane i keeps S_i = (y1_i, ig_i)
        float fy1 = y1, fig = ig, fx1 = err_prev, my_y1 = y1, my_ig = ig;
#pragma unroll
        for (int i = 1; i < K; i++) {
            const float t0 = fmaf(fx1, kc.b0, fy1 * kc.a0);
            fy1 = (0.0f + t0) + t1v[i - 1]; fx1 = ev[i - 1];
            fig = fmaf(ev[i - 1], kc.ktsi, fig);
            my_y1 = (i == j) ? fy1 : my_y1; my_ig = (i == j) ? fig : my_ig;
        }
        // the integrator moves < 4e-6 per sample: its clamp acted nowhere in the span iff both ends are well inside
        const bool integ_clamped = !((fabsf(ig) <= 0.99f) && (fabsf(fig) <= 0.99f));
        const float Fj = fmaf(clampf(my_ig + my_y1 * kc.c001, -1.0f, 1.0f), kc.m100, kc.m19000);
        const bool ok_j = (f32_bits(Fj) == f32_bits(Fmine)) && (j < rem);
        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok_j || j == 0);
        int m = __builtin_ctz(~((unsigned int)(okm >> (g * K)) & ((1u << K) - 1u)) | (1u << K));   // first invalid sample (K = none)
        float nx, ny, ni, ne, nt;
        if (__builtin_amdgcn_ballot_w64(active && integ_clamped) != 0ull) {
            // a saturated integrator (never in lock): verify with the exact clamps, predicated
            // (from the registers (D) already holds, unrolled: a loop on its rail — a pilot beyond the NCO's range, a dead channel's NaN —
            // takes this path on every span, and the LDS reads of the rolled form were seven round trips a span)
            float x1 = err_prev, yy1 = y1, ig2 = ig;
            bool valid = true;
            m = 1;
#pragma unroll
            for (int i = 1; i < K; i++) {
                const float ei = ev[i - 1];
                const float t0 = fmaf(x1, kc.b0, yy1 * kc.a0);
                const float ny1 = (0.0f + t0) + t1v[i - 1];
                const float ni1 = clampf(fmaf(ei, kc.ktsi, ig2), -1.0f, 1.0f);
                const float Fi = fmaf(clampf((ni1 + ny1 * kc.c001) * 1.0f, -1.0f, 1.0f), kc.m100, kc.m19000);
                valid = valid && (f32_bits(Fi) == f32_bits(seq ? Fq[i] : F)) && (i < rem);
                x1 = valid ? ei : x1; yy1 = valid ? ny1 : yy1; ig2 = valid ? ni1 : ig2; m = valid ? i + 1 : m;
            }
            nx = x1; ny = yy1; ni = ig2; ne = ex[g][m]; nt = dts[g][(pos + m - 1) & (RING - 1)];
        } else {
            // resume state S_{m-1}, err_{m-1}, tph_{m-1}: held by lane m-1 of the channel
            const int src = (g * K + m - 1) * 4;
            ny = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(my_y1)));
            ni = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(my_ig)));
            ne = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(e)));
            nt = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(mine)));
            nx = ex[g][m - 1];                               // err_{m-2} (ex[g][0] = err_prev)
        }
        if (seq) {
            few = __builtin_amdgcn_ballot_w64(active && m == 1 && rem > 1) != 0ull;
            const float Fprev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Fmine), 0x111, 0xf, 0xf, true));      // row_shr:1 (a channel's lanes share a row)
            const unsigned long long chg = __builtin_amdgcn_ballot_w64(active && j > 0 && f32_bits(Fmine) != f32_bits(Fprev));
            const int nchg = __builtin_popcount((unsigned int)(chg >> (g * K)) & ((1u << K) - 1u));      // this channel's words changed that often in the span
            changes = (__builtin_amdgcn_ballot_w64(nchg >= 1) != 0ull) + (__builtin_amdgcn_ballot_w64(nchg >= 2) != 0ull) +
                      (__builtin_amdgcn_ballot_w64(nchg >= 3) != 0ull) + (__builtin_amdgcn_ballot_w64(nchg >= 4) != 0ull);
        }
        if (active) {
            lx1 = nx; ly1 = ny; integ = ni; err_prev = ne; tph_prev = nt; pos += m;
        }
        ex[g][0] = err_prev;
    };
    for (int q = q_resume; q < chunks; q++) {
        const int cend = (q + 1) * CH;
        int spans = 0;
        bool serial = seq_left > 0;
        if (!serial) {
            // Which form: "F stays put" until a chunk has taken kPllSeqSpans spans (1.8 x the fewest: on some channel that form commits fewer
            // than K / 1.8 samples a span, what a span of the sequence form costs) — the rest of the chunk and the following chunks in the
            // sequence form, until over a whole chunk the other form would have needed clearly fewer than 1.8 spans per span of this one
            // (1 + the changes of the guessed words).  One scalar compare a span in the constant-word form.
            // (Round 3's rule — a whole chunk of one-sample spans, then serial chunks with an exponential back-off and a speculative chunk
            // to probe — spent a third of an unlocked wavefront's time in the probes.)
            int seq_spans = 0, chg_sum = 0;
            while (__builtin_amdgcn_ballot_w64(pos < cend) != 0ull) {
                bool few = false;
                int changes = 0;
                spans++;
                const bool sq = seq_form;
                span(sq, cend, few, changes);          // (one call site: one copy of the body)
                if (!sq) {
#ifndef FMD_PLL_NOSEQ
                    if (spans >= kPllSeqSpans) seq_form = true;
#endif
                } else {
                    seq_spans++;
                    chg_sum += changes;
                    stuck = few ? stuck + 1 : 0;
                    if (stuck >= kPllStuckSpans) break;
                }
            }
            n_seq_spans += (unsigned long long)seq_spans;
            if (seq_form && stuck >= kPllStuckSpans) {
                // one sample a span in the sequence form too (a guess pass that cannot predict the words: not seen on any signal of the tests;
                // NaN input does it): the rest of this chunk and the next ones serially, doubling each time in a row
                serial = true;
                seq_left = 1 + (hold ? hold : 1); hold = hold ? (2 * hold < kSlowHoldMax ? 2 * hold : kSlowHoldMax) : 1; stuck = 0;
            } else {
                if (seq_form && seq_spans == spans && 10 * chg_sum < 4 * seq_spans) seq_form = false;
                if (spans <= kPllSlowSpans) hold = 0;
            }
        }
        if (serial) {
            // ---- neither form commits: the plain serial iteration, computed identically by the 16 lanes of a channel ----
            seq_left--; n_seq++;
            PllState S{lx1, ly1, integ, err_prev, tph_prev};
            while (__builtin_amdgcn_ballot_w64(pos < cend) != 0ull) {
                if (pos < cend) {
                    const float2 x = xin[g][pos & (RING - 1)];
                    dts[g][pos & (RING - 1)] = pll_step(S, gain * x.x, gain * x.y, k);
                    pos++;
                }
            }
            lx1 = S.lx1; ly1 = S.ly1; integ = S.integ; err_prev = S.err; tph_prev = S.tph;
            ex[g][0] = err_prev;
        }
        // chunk q is final in every channel: drain it, refill its half of the ring with chunk q+2, prefetch chunk q+3
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {   // results of chunk q
            const float* row = &dts[g][(q & 1) * CH + 4 * j];
            float* o = pll_dt + (size_t)cs * n + (size_t)q * CH + 4 * j;
#define FMD_PLL_DRAIN(r) if constexpr (r < NST) { const float4 v_ = *reinterpret_cast<const float4*>(row + 4 * K * r); if (live) *reinterpret_cast<float4*>(o + 4 * K * r) = v_; }
            FMD_PLL_EACH(FMD_PLL_DRAIN)
        }
        FMD_PLL_STASH(q + 2)
        FMD_PLL_FETCH(q + 3)
        n_spans += spans;
    }
    }
    if (live && j == 0) {
        st(state, S_AGC_PILOT_GAIN, d.C, c) = gain;
        st(state, S_PLL_X1, d.C, c) = lx1; st(state, S_PLL_Y1, d.C, c) = ly1;
        st(state, S_PLL_INT, d.C, c) = integ; st(state, S_PLL_ERR, d.C, c) = err_prev; st(state, S_PLL_T, d.C, c) = tph_prev;
    }
    // which body the next block's launch runs for these stations: the sequence-capable one after a block that ended out of lock (or went serial)
    // (back to the other body only after a block that never left the constant-word form: a loop that wanders in and out of lock keeps this one)
    // (per station, so that the 16- and the 8-lane kernel can follow one another: fmd_api.cpp picks the lane count by what is out of lock)
    if (hint) {
        const unsigned int out = (n_seq != 0ull || n_seq_spans != 0ull || handed_over) ? 1u : 0u;
        if (live && j == 0) hint[c] = out;
        // [C]: the newest launch in which a wavefront spent a quarter of the block or more out of lock, [C + 1]: the newest launch that has run — the
        // host watches the distance (a chunk or two in the other form is what any loop does now and then: not counted)
        const bool heavy = 4ull * n_seq >= (unsigned long long)chunks || 4ull * n_seq_spans >= (unsigned long long)(n_spans > 0 ? n_spans : 1);
        if (heavy && lane == 0) atomicMax(hint + d.C, launch_no);
        if (blockIdx.x == 0 && lane == 0) atomicMax(hint + d.C + 1, launch_no);
    }
    if (chain) {   // publish: the state stores above, then the sequence number
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&chain[blockIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (spec_stats) {
        // per wavefront: chunks / chunks run serially / spans redone with the reference forms; per channel: spans, samples
        if (lane == 0) { atomicAdd(&spec_stats[0], (unsigned long long)chunks); atomicAdd(&spec_stats[1], n_seq); atomicAdd(&spec_stats[2], n_exact); atomicAdd(&spec_stats[5], n_seq_spans); }
        // per channel: the spans its wavefront ran, and the samples they covered (every sample of a speculative chunk)
        if (live && j == 0) { atomicAdd(&spec_stats[3], (unsigned long long)n_spans); atomicAdd(&spec_stats[4], ((unsigned long long)chunks - n_seq) * CH); }
        // shader-clock cycles and 100 MHz real-time ticks this wavefront ran: their ratio is the core clock the power
        // management granted while the other stages' kernels ran beside it (DESIGN.md "Clocks")
        if (lane == 0 && blockIdx.x == 0) { atomicAdd(&spec_stats[6], __builtin_readcyclecounter() - clk0); atomicAdd(&spec_stats[7], __builtin_amdgcn_s_memrealtime() - rt0); }
    }
}

// The kernel: the LDS, and pilot_pll_body — one prologue and epilogue around two chunk loops, chosen per wavefront by its stations' flags
// (Buffers::pll_hint, written by the previous block's launch).  Two loops rather than one with both forms: one span body holding both cost
// the all-locked batch 5-18 % (branches and register moves in the constant-word span, and the instruction cache when single spans strayed
// into the other form's copy); a wavefront runs one loop for a whole block, so its code stays resident, and the all-locked batch keeps
// round 3's loop (same-box A/B against a build without the second loop: +0.9 % on the step).
template <int K>
__global__ __launch_bounds__(kWave) void k_pilot_pll(Dims d, const float2* __restrict__ pilot, float* __restrict__ pll_dt,
                                                     float* __restrict__ state, LoopCoeffs k, int power_field,
                                                     unsigned long long* __restrict__ spec_stats,
                                                     unsigned int* __restrict__ chain, unsigned int seq, unsigned int* __restrict__ hint, unsigned int launch_no) {
    constexpr int G = kWave / K, RING = kPllRing;
    __shared__ __attribute__((aligned(16))) float2 xin[G][RING];   // pilot samples, ring by (sample index & 255)
    __shared__ __attribute__((aligned(16))) float dts[G][RING];    // results, same ring
    __shared__ __attribute__((aligned(16))) float ex[G][K + 4];    // [0] = err_prev, [1 + i] = err_i of the current span
    __shared__ __attribute__((aligned(16))) float e1x[G][K + 4];   // fma(err_i, b1, 0)
    __shared__ __attribute__((aligned(16))) float fsq[G][K + 4];   // out of lock: the guessed frequency word of every sample of the span
    pilot_pll_body<K>(d, pilot, pll_dt, state, k, power_field, spec_stats, chain, seq, hint, launch_no, xin, dts, ex, e1x, fsq);
}

// ===============================================================================================================
// k_pilot_pll_pairs — the LOW-WORK variant of the pilot PLL kernel, used for large batches (> kPllTimeParallelMaxChannels).
// The time-parallel kernel above spends ~2.6x the VALU instructions of this one to cut the latency of a lone wavefront in
// half; once the batch is large enough that the chip's VALU throughput, not a lone wavefront's latency, bounds the step
// (measured: from ~8192 channels per GPU), the cheaper kernel wins.  Same arithmetic, same results.
// ===============================================================================================================
// ---------------------------------------------------------------------------------------------------------------
// Speculative form of the same iteration for a loop that is in lock.  A lone wavefront per SIMD issues one instruction
// every ~4 cycles and waits ~6 cycles on a dependent result, so the loop's duration is max(4 x instructions, 6 x chain
// length).  pll_step (the reference iteration) is a 78-operation err -> err chain; pll_step_pair cuts the chain to ~45
// and the instruction count to ~68 per sample:
//   * the integrator / control clamps are skipped (shown not to bind for the whole chunk by pll_chunk_precheck),
//   * x - round_half_away(x) becomes x - rndne(x) (equal unless x is an exact tie, detected from the chebyshev
//     argument: wrapped phase == +-0.5  <=>  z - 0.25 == 0),
//   * the phase detector is atan2f's first range with an unscaled division (fmd_math.h div_unscaled), valid for
//     x in [2^-28, ~2e8) and 2^-29 <= |y/x| < 7/16, i.e. (y/x)^2 in [2^-58, 49/256) — both windows are 0x1bc40000 wide
//     in the float's bit pattern, so one unsigned max tracks both,
//   * TWO LANES PER CHANNEL: the iteration contains two pairs of structurally identical, mutually independent
//     polynomial evaluations (chebyshev sine of the phase and of phase + 1/4; the odd and even halves s1 / s2 of the
//     arctangent series).  The even lane of a pair evaluates the first of each, the odd lane the second, with the same
//     instructions and per-lane constants, and they exchange the results with DPP quad permutes.  Everything else is
//     computed redundantly (identically) by both lanes.  A wavefront therefore carries 32 channels.
// Validity is accumulated in VALU registers only (a v_cmp -> SALU round trip stalls an in-order wave ~18 cycles) and
// tested once per 16-sample chunk; a chunk with any invalid lane is replayed with pll_step.  Constants live in VGPRs:
// a 32-bit literal in the instruction stream costs a lone wave ~2.7 extra cycles per instruction.
// ---------------------------------------------------------------------------------------------------------------
struct PairConsts {
    float b0, a0, b1, c001, ktsi, m100, m19000, ts, q25 /* odd lane 1/4, even lane -0 */, mq25, c5, c4, c3, c2, c1, c0;
    float k0, k1, k2, k3, k4, k5;   // arctangent series, this lane's half: even lane a10,a8,a6,a4,a2,a0 (s1); odd lane 0,a9,a7,a5,a3,a1 (s2)
    uint32_t xlo, zlo;
    bool odd;
};
__device__ __forceinline__ PairConsts make_pair_consts(const LoopCoeffs& k, bool odd) {
    PairConsts c;
    FMD_OPAQUE_F(c.b0, k.pll_b0) FMD_OPAQUE_F(c.a0, k.pll_a0) FMD_OPAQUE_F(c.b1, k.pll_b1) FMD_OPAQUE_F(c.c001, 0.01f)
    FMD_OPAQUE_F(c.ktsi, 0.1f * (1.0f / 128000.0f)) FMD_OPAQUE_F(c.m100, -100.0f) FMD_OPAQUE_F(c.m19000, -19000.0f)
    FMD_OPAQUE_F(c.ts, 1.0f / 128000.0f) FMD_OPAQUE_F(c.q25, odd ? 0.25f : -0.0f) FMD_OPAQUE_F(c.mq25, -0.25f)
    FMD_OPAQUE_F(c.c5, 3.20396066f) FMD_OPAQUE_F(c.c4, -14.07150173f) FMD_OPAQUE_F(c.c3, 38.50016403f)
    FMD_OPAQUE_F(c.c2, -67.07687378f) FMD_OPAQUE_F(c.c1, 64.83583069f) FMD_OPAQUE_F(c.c0, -25.13274193f)
    FMD_OPAQUE_F(c.k0, odd ? 0.0f : bits_f32(0x3c8569d7u))
    FMD_OPAQUE_F(c.k1, odd ? bits_f32(0xbd15a221u) : bits_f32(0x3d4bda59u))
    FMD_OPAQUE_F(c.k2, odd ? bits_f32(0xbd6ef16bu) : bits_f32(0x3d886b35u))
    FMD_OPAQUE_F(c.k3, odd ? bits_f32(0xbd9d8795u) : bits_f32(0x3dba2e6eu))
    FMD_OPAQUE_F(c.k4, odd ? bits_f32(0xbde38e38u) : bits_f32(0x3e124925u))
    FMD_OPAQUE_F(c.k5, odd ? bits_f32(0xbe4ccccdu) : bits_f32(0x3eaaaaabu))
    FMD_OPAQUE_U(c.xlo, 0x31800000u) FMD_OPAQUE_U(c.zlo, 0x22800000u)
    c.odd = odd;
    return c;
}

struct PairChecks { float tie_min; uint32_t range_max; };
static constexpr uint32_t kPairRangeWindow = 0x1bc40000u;   // bits(49/256) - bits(2^-58)

// DPP quad permutes between the two lanes of a channel pair (lanes 2j, 2j+1)
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
static constexpr int kDppEven = 0xA0;   // quad_perm [0,0,2,2]: both lanes read the even lane
static constexpr int kDppOdd = 0xF5;    // quad_perm [1,1,3,3]: both lanes read the odd lane
static constexpr int kDppSwap = 0xB1;   // quad_perm [1,0,3,2]: each lane reads its partner

// atan2f(y, x) for x in [2^-28, ~2e8) and 2^-29 <= |y/x| < 7/16 (a locked loop's phase error): the published algorithm
// reduces to t - t (s1 + s2) with t = y / x (its first range, which is odd-symmetric, so no quadrant or sign selects), and
// the division needs neither operand scaling nor special-value fix-up.  Both lanes of a pair hold the same (y, x); the even
// lane evaluates s1 = z (a0 + w (a2 + w (a4 + w (a6 + w (a8 + w a10))))), the odd lane s2 = w (a1 + w (a3 + w (a5 + w (a7 + w a9))))
// (its first level is a9 + w 0 = a9 exactly), then each adds its partner's half (IEEE addition commutes).
__device__ __forceinline__ float atan2f_pair(float y, float x, const PairConsts& c, PairChecks& ck) {
    const float t = div_unscaled(y, x);
    const float z = t * t;
    const float w = z * z;
    ck.range_max = max(max(ck.range_max, f32_bits(x) - c.xlo), f32_bits(z) - c.zlo);
    float s = c.k1 + w * c.k0;
    s = c.k2 + w * s;
    s = c.k3 + w * s;
    s = c.k4 + w * s;
    s = c.k5 + w * s;
    s = (c.odd ? w : z) * s;
    const float sum = s + dpp_quad<kDppSwap>(s);
    return t - t * sum;
}

// Holds for the whole chunk if it holds at its start, given that every err the chunk produces is < 0.42 in magnitude
// (implied by the range check): the loop filter is a convex combination (|lpf| <= max of its inputs), the integrator
// moves by < 4e-6 per sample, so |integ| <= 0.9001, |integ + 0.01 lpf| <= 0.95 < 1 and |t + Ts freq| <= 0.65 < 1.5.
__device__ __forceinline__ bool pll_chunk_precheck(const PllState& s, const LoopCoeffs& k) {
    const bool convex = (k.pll_b0 >= 0.0f) && (k.pll_b1 >= 0.0f) && (k.pll_a0 >= 0.0f) && ((k.pll_b0 + k.pll_b1) + k.pll_a0 <= 1.0001f);
    return convex && (fabsf(s.integ) <= 0.9f) && (fabsf(s.lx1) <= 4.0f) && (fabsf(s.ly1) <= 4.0f) && (fabsf(s.err) <= 4.0f) && (fabsf(s.tph) <= 0.5f);
}

__device__ __forceinline__ float pll_step_pair(PllState& s, float p, float q, const PairConsts& c, PairChecks& ck) {
    const float t0 = fmaf(s.lx1, c.b0, s.ly1 * c.a0);
    const float t1 = fmaf(s.err, c.b1, 0.0f);
    const float lpf = (0.0f + t0) + t1;
    s.lx1 = s.err; s.ly1 = lpf;
    const float P = lpf * c.c001;
    s.integ = fmaf(s.err, c.ktsi, s.integ);
    const float PI_error = s.integ + P;
    const float freq = fmaf(PI_error, c.m100, c.m19000);
    const float yy = fmaf(freq, c.ts, s.tph);
    s.tph = yy - rintf(yy);
    // chebyshev_sine (scalar association): the odd lane of the pair takes sin(2 pi wrap(t + 1/4)); the even lane runs the
    // same three instructions with -0 in place of 1/4, which leave t unchanged bit for bit (t + -0 = t, |t| <= 1/2 so
    // rndne(t) = +-0), i.e. sin(2 pi t) — no select on the dependency chain
    const float dc = s.tph + c.q25;
    const float xr = dc - rintf(dc);
    const float z = xr * xr;
    float poly = fmaf(c.c5, z, c.c4);
    poly = fmaf(poly, z, c.c3);
    poly = fmaf(poly, z, c.c2);
    poly = fmaf(poly, z, c.c1);
    poly = fmaf(poly, z, c.c0);
    const float zq = z + c.mq25;
    const float sn = (zq * xr) * poly;
    ck.tie_min = fminf(ck.tie_min, fabsf(zq));   // wrapped phase == +-1/2 <=> zq == 0: the rndne shortcut was not exact
    const float ps = dpp_quad<kDppEven>(sn), pc = dpp_quad<kDppOdd>(sn);
    const float res_im = fmaf(ps, p, q * pc);
    const float res_re = fmaf(p, pc, -(q * ps));
    s.err = atan2f_pair(res_im, res_re, c, ck);
    return s.tph;
}

static constexpr int kPairSlowHoldMax = 64;   // longest run of general-form chunks between two speculation attempts

// ---------------------------------------------------------------------------------------------------------------
// 16-sample chunk staging for the two-wave pilot PLL kernel: 32 channels per workgroup (two lanes per channel in the
// recurrence wave), 4 float4 registers per chunk in the mover wave.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int kCh16 = 16;
static constexpr int kRow16 = kCh16 + 2;   // float2 row stride (144 B) of a transposed 16-sample cf32 chunk
static constexpr int kPllCh = 32;          // channels per k_pilot_pll_pairs workgroup
struct Chunk16 { float4 v0, v1, v2, v3; };
#define FMD_FOR4(X) X(0) X(1) X(2) X(3)
__device__ __forceinline__ Chunk16 chunk16_load(const float2* __restrict__ base, int n, int c0, int C, int t0) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 3, col = lane & 7;
    Chunk16 r;
#define FMD_LD4(k) { int ch = c0 + 8 * k + row; ch = ch < C ? ch : C - 1; \
                     r.v##k = *reinterpret_cast<const float4*>(base + (size_t)ch * n + t0 + 2 * col); }
    FMD_FOR4(FMD_LD4)
#undef FMD_LD4
    return r;
}
__device__ __forceinline__ void chunk16_store(const Chunk16& r, float2* lds) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 3, col = lane & 7;
#define FMD_ST4(k) *reinterpret_cast<float4*>(lds + (8 * k + row) * kRow16 + 2 * col) = r.v##k;
    FMD_FOR4(FMD_ST4)
#undef FMD_ST4
}
// drain 16 f32 results per channel, stored compactly at the start of each row of a chunk buffer, to out[C][n] at t0
__device__ __forceinline__ void chunk16_flush_f(const float2* lds, float* __restrict__ out, int n, int c0, int C, int t0) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 2, col = lane & 3;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = 16 * k + row, ch = c0 + r;
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(lds + r * kRow16) + 4 * col);
        if (ch < C) *reinterpret_cast<float4*>(out + (size_t)ch * n + t0 + 4 * col) = v;
    }
}

// Workgroup = two wavefronts for 32 channels.  Wave 0 runs the recurrence (two lanes per channel, see pll_step_pair) and
// touches only LDS; wave 1 (the mover) stages the next chunks HBM -> registers -> LDS and drains finished pll_dt chunks
// LDS -> HBM.  A lone wave is bound by its own instruction issue, and a vector-memory instruction costs it tens of cycles
// — hundreds when other stages' kernels keep the CU's memory pipeline busy — so the memory instructions are given to a
// sibling on another SIMD; the two meet at one barrier per 16-sample chunk.
// The workgroup is kept SMALL on purpose (9 KB of LDS, results written in place over consumed input; well under 150
// VGPRs per wave): while the FIR stages' kernels fill every CU, a serial-stage workgroup that needs 52 KB and 2 x 256
// registers waits >100 us for a hole (measured, tools/gap_probe.hip), one that fits the hole a retiring FIR workgroup
// leaves starts at once.
__global__ __launch_bounds__(2 * kWave) void k_pilot_pll_pairs(Dims d, const float2* __restrict__ pilot, float* __restrict__ pll_dt,
                                                         float* __restrict__ state, LoopCoeffs k, int power_field,
                                                         unsigned long long* __restrict__ spec_stats, unsigned int* __restrict__ hint, unsigned int launch_no) {
    __shared__ __attribute__((aligned(16))) float2 ring[2][kPllCh * kRow16];
    const bool mover = threadIdx.x >= kWave;   // wave-uniform
    const int lane = threadIdx.x & (kWave - 1), c0 = blockIdx.x * kPllCh;
    const int n = d.n_fm_out, chunks = n / kCh16;

    if (mover) {
        // iteration ch (between barriers ch and ch+1): drain the results of chunk ch-1 from its slot, refill that slot with
        // chunk ch+1 (loaded three iterations ago), issue the loads of chunk ch+4
        auto load_or_last = [&](int ch) { return chunk16_load(pilot, n, c0, d.C, (ch < chunks ? ch : chunks - 1) * kCh16); };
        Chunk16 r0 = load_or_last(0);
        chunk16_store(r0, ring[0]);
        Chunk16 ra = load_or_last(1), rb = load_or_last(2), rc = load_or_last(3);
        auto move_chunk = [&](int ch, Chunk16& regs) {
            float2* slot = ring[(ch + 1) & 1];
            if (ch > 0) chunk16_flush_f(slot, pll_dt, n, c0, d.C, (ch - 1) * kCh16);
            if (ch + 1 < chunks) chunk16_store(regs, slot);
            if (ch + 4 < chunks) regs = load_or_last(ch + 4);
        };
        for (int ch = 0; ch < chunks; ch += 3) {
            __syncthreads();
            move_chunk(ch, ra);
            if (ch + 1 < chunks) { __syncthreads(); move_chunk(ch + 1, rb); }
            if (ch + 2 < chunks) { __syncthreads(); move_chunk(ch + 2, rc); }
        }
        __syncthreads();
        chunk16_flush_f(ring[(chunks - 1) & 1], pll_dt, n, c0, d.C, (chunks - 1) * kCh16);
        return;
    }

    __builtin_amdgcn_s_setprio(3);  // latency-bound recurrence: win issue arbitration against co-resident FIR waves
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    const int c = c0 + (lane >> 1);            // lanes 2j and 2j+1 both carry channel c0 + j
    const bool odd = (lane & 1) != 0;
    const bool live = c < d.C;
    const int cs = live ? c : d.C - 1;
    // AGC_Filter::process (agc.h:12-19) as compiled: target_gain = sqrt((target/sum) * N)
    float gain = st(state, S_AGC_PILOT_GAIN, d.C, cs);
    {
        const float sum = st(state, power_field, d.C, cs);
        const float target_gain = sqrtf((1.0f / sum) * (float)n);
        gain = fmaf(target_gain - gain, 0.2f, gain);
    }
    PllState S;
    S.lx1 = st(state, S_PLL_X1, d.C, cs); S.ly1 = st(state, S_PLL_Y1, d.C, cs);
    S.integ = st(state, S_PLL_INT, d.C, cs); S.err = st(state, S_PLL_ERR, d.C, cs); S.tph = st(state, S_PLL_T, d.C, cs);
    const PairConsts kc = make_pair_consts(k, odd);
    // a failed speculative chunk is replayed with the general forms; consecutive failures (a loop out of lock) back
    // off exponentially so an unlocked wavefront pays at most a few percent for its attempts
    int slow_left = 0, hold = 0, n_replayed = 0, n_general = 0;
    for (int ch = 0; ch < chunks; ch++) {
        // here: ring[ch & 1] holds chunk ch; the other slot holds the results of chunk ch - 1, which the mover now drains
        __syncthreads();
        float2* row = ring[ch & 1] + (lane >> 1) * kRow16;
        float* dto = reinterpret_cast<float*>(row);   // result t goes to float t of the row: x[t/2] has been consumed by then
        bool done = false;
        if (slow_left == 0) {
            PllState s = S;
            PairChecks ck{1.0f, 0u};
            float2 xs[kCh16];                         // the chunk is read before any result is written over it
#pragma unroll
            for (int t = 0; t < kCh16; t++) xs[t] = row[t];
            float dts[kCh16];
#pragma unroll
            for (int t = 0; t < kCh16; t++) dts[t] = pll_step_pair(s, gain * xs[t].x, gain * xs[t].y, kc, ck);
            const bool ok = pll_chunk_precheck(S, k) && (ck.tie_min != 0.0f) && (ck.range_max < kPairRangeWindow);
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
                S = s; done = true; hold = 0;
                if (!odd) {
#pragma unroll
                    for (int t = 0; t < kCh16; t += 4) *reinterpret_cast<float4*>(dto + t) = make_float4(dts[t], dts[t + 1], dts[t + 2], dts[t + 3]);
                }
            } else { slow_left = hold; hold = hold ? (2 * hold < kPairSlowHoldMax ? 2 * hold : kPairSlowHoldMax) : 1; n_replayed++; }
        } else {
            slow_left--;
        }
        if (!done) {   // general iteration, computed identically by both lanes of a pair (same address, same value)
            n_general++;
            float2 y = row[0];
            for (int t = 0; t < kCh16; t++) {
                const float2 yn = row[t + 1 < kCh16 ? t + 1 : t];
                dto[t] = pll_step(S, gain * y.x, gain * y.y, k);
                y = yn;
            }
        }
        // The results must have landed in LDS before the mover reads them after the next barrier.  The compiler's own
        // wait in front of the loop-header s_barrier went missing on this back edge (seen in the ISA and as stale last
        // samples at 4096 channels), so it is spelled out.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (live && !odd) {
        st(state, S_AGC_PILOT_GAIN, d.C, c) = gain;
        st(state, S_PLL_X1, d.C, c) = S.lx1; st(state, S_PLL_Y1, d.C, c) = S.ly1;
        st(state, S_PLL_INT, d.C, c) = S.integ; st(state, S_PLL_ERR, d.C, c) = S.err; st(state, S_PLL_T, d.C, c) = S.tph;
    }
    if (hint) {
        // Buffers::pll_hint as k_pilot_pll keeps it: a wavefront that ran a quarter of the block or more with the general iteration has a loop out
        // of lock — this kernel has no cheaper way through it, the time-parallel kernel's sequence form has, and the host (fmd_api.cpp) launches
        // that one for the next blocks while the counter moves
        const bool heavy = 4 * n_general >= chunks;
        if (live && !odd) hint[c] = heavy ? 1u : 0u;
        if (heavy && lane == 0) atomicMax(hint + d.C, launch_no);
        if (blockIdx.x == 0 && lane == 0) atomicMax(hint + d.C + 1, launch_no);
    }
    if (lane == 0 && spec_stats) {   // same slots as k_pilot_pll: chunks (16 samples here) / run with the general iteration / replayed
        atomicAdd(&spec_stats[0], (unsigned long long)chunks);
        atomicAdd(&spec_stats[1], (unsigned long long)n_general);
        atomicAdd(&spec_stats[2], (unsigned long long)n_replayed);
        // shader-clock cycles and 100 MHz real-time ticks this wavefront ran: their ratio is the core clock the power
        // management granted while the other stages' kernels ran beside it (DESIGN.md "Clocks")
        if (blockIdx.x == 0) { atomicAdd(&spec_stats[6], __builtin_readcyclecounter() - clk0); atomicAdd(&spec_stats[7], __builtin_amdgcn_s_memrealtime() - rt0); }
    }
}

// =============================================================================================
// k_extract — reference ExtractComponents (broadcast_fm_demod.cpp:463-536) + MixAudio (:549-585):
//   apply_harmonic_pll_avx (apply_harmonic_pll.cpp:88-139) h=2 (+ L-R phase offset) and h=3, fused into
//   PolyphaseDownsampler<cf32> 4x128 (L+R: real rail only; L-R: imaginary rail, plus the real rail of every 10th
//   output for the phase estimate :496-510) and 8x128 (RDS); stereo mix.
// One workgroup = one channel x TA audio samples (4 TA fm_out samples, TA/2 RDS samples), TA threads.
// The mixed signals are staged in LDS split into decimation phases; every FIR thread produces FOUR consecutive
// outputs from one sliding register window (ds_read_b128), so an LDS byte feeds ~4 FMAs instead of 1.
// =============================================================================================
template <int TA>
struct ExtractGeom {
    static constexpr int XS = 4 * TA + 124;      // fm_out samples staged
    static constexpr int P4 = TA + 40;           // phase stride (floats) for the decimate-by-4 signals: == 8 (mod 32), multiple of 4
    static constexpr int P8 = TA / 2 + 20;       // phase stride for the decimate-by-8 signal: == 20 (mod 32), multiple of 4
    static constexpr int NQ = TA / 4;            // threads per FIR job (each makes 4 outputs)
    static constexpr int NEST = TA / 10 + 2;     // upper bound of phase-estimate outputs per tile
    static_assert(P4 % 4 == 0 && P8 % 4 == 0 && P4 >= TA + 31 && P8 >= TA / 2 + 15, "phase strides");
};

__device__ __forceinline__ float2 harmonic_mix(float2 x, float dt, float harmonic, float off, float off_cos) {
    float s = fmaf(dt, harmonic, off);
    float c = fmaf(dt, harmonic, off_cos);
    s = s - rintf(s);
    c = c - rintf(c);
    const float pc = cheb_sine_vector(c), ps = cheb_sine_vector(s);
    return make_float2(fmaf(pc, x.x, -(x.y * ps)), fmaf(pc, x.y, x.x * ps));
}

// Four consecutive outputs of a decimate-by-4, 128-tap FIR on one rail.  ph: 4 phase arrays of stride P, window of
// output i starts at phase index i.  Accumulation order per output = c32_f32_cum_mul_avx: lane p sums taps 4jj+p in
// increasing jj, then (l0+l2)+(l1+l3).
template <typename TapFn>
__device__ __forceinline__ void fir4_quad(const float* __restrict__ ph, int P, int u, TapFn tap, float out[4]) {
    float acc[4][4];
#pragma unroll
    for (int v = 0; v < 4; v++) { acc[v][0] = acc[v][1] = acc[v][2] = acc[v][3] = 0.0f; }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        float w[36];
        const float4* src = reinterpret_cast<const float4*>(ph + p * P + 4 * u);
#pragma unroll
        for (int k = 0; k < 9; k++) { const float4 t = src[k]; w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w; }
#pragma unroll
        for (int jj = 0; jj < 32; jj++) {
            const float b = tap(4 * jj + p);
#pragma unroll
            for (int v = 0; v < 4; v++) acc[v][p] = fmaf(w[v + jj], b, acc[v][p]);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; v++) out[v] = (acc[v][0] + acc[v][2]) + (acc[v][1] + acc[v][3]);
}

// Four consecutive outputs of a decimate-by-8, 128-tap FIR on one rail: 8 phase arrays; lane (n & 3) sums taps n = 8jj+p
// in increasing n, so phases p and p+4 feed the same accumulator alternately.
template <typename TapFn>
__device__ __forceinline__ void fir8_quad(const float* __restrict__ ph, int P, int u, TapFn tap, float out[4]) {
    float acc[4][4];
#pragma unroll
    for (int v = 0; v < 4; v++) { acc[v][0] = acc[v][1] = acc[v][2] = acc[v][3] = 0.0f; }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        float wa[20], wb[20];
        const float4* sa = reinterpret_cast<const float4*>(ph + p * P + 4 * u);
        const float4* sb = reinterpret_cast<const float4*>(ph + (p + 4) * P + 4 * u);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float4 t = sa[k]; wa[4 * k] = t.x; wa[4 * k + 1] = t.y; wa[4 * k + 2] = t.z; wa[4 * k + 3] = t.w;
            const float4 r = sb[k]; wb[4 * k] = r.x; wb[4 * k + 1] = r.y; wb[4 * k + 2] = r.z; wb[4 * k + 3] = r.w;
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const float ba = tap(8 * jj + p), bb = tap(8 * jj + p + 4);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                acc[v][p] = fmaf(wa[v + jj], ba, acc[v][p]);
                acc[v][p] = fmaf(wb[v + jj], bb, acc[v][p]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 4; v++) out[v] = (acc[v][0] + acc[v][2]) + (acc[v][1] + acc[v][3]);
}

constexpr int kLmrInlineMax = 512;   // estimates per block (n_audio / 10) up to which k_extract integrates the L-R phase itself

// reference ExtractComponents :511-516 — avg = sum / n; phase = fmod(phase + 0.1 avg, 2 pi)
__device__ __forceinline__ float lmr_phase_finish(float sum, int n_est, float cur) {
    const float avg = sum / (float)n_est;
    const float acc = fmaf(avg, 0.1f, cur);
    return fmodf(acc, bits_f32(kTwoPiBits));
}
// sum over the 64 lanes of a wavefront, the same value in every lane
// (the xor butterfly 32, 16, 8, 4, 2, 1 of six __shfl_xor steps — same partner in every step, so the same sum bit for bit — without their
//  six LDS round trips (ds_bpermute: ~130 cycles each on a lone dependent chain): lane-swap and DPP moves)
__device__ __forceinline__ float wave_sum_f32(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);       // halves of the wavefront change places
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);                                                      // v[i] + v[i ^ 32]
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);            // odd rows of 16 <-> even rows
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);                                                      // + v[i ^ 16]
    v += dpp0<0x128>(v);                                                                                    // row_ror:8: + v[i ^ 8]
    {   // + v[i ^ 4]: row_shl:4 into lanes 0-3 and 8-11 of a row, row_shr:4 into the others
        const int up = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, true);
        v += __int_as_float(__builtin_amdgcn_update_dpp(up, __float_as_int(v), 0x114, 0xf, 0xA, false));
    }
    v += dpp0<0x4E>(v);                                                                                     // quad_perm [2,3,0,1]: + v[i ^ 2]
    v += dpp0<0xB1>(v);                                                                                     // quad_perm [1,0,3,2]: + v[i ^ 1]
    return v;
}

template <int TA, bool FAST = false>
__global__ __launch_bounds__(TA) void k_extract(Dims d, const float2* __restrict__ fm_out_iq, const float* __restrict__ pll_dt,
                                                const float2* __restrict__ iq_tail_in, const float* __restrict__ dt_tail_in,
                                                float2* __restrict__ iq_tail_out, float* __restrict__ dt_tail_out,
                                                const float* __restrict__ b_lpr, const float* __restrict__ b_lmr, RdsTaps rds_taps,
                                                const float* __restrict__ mixctl, float* __restrict__ state,
                                                float* __restrict__ audio, float2* __restrict__ rds, float* __restrict__ lmr_est,
                                                float* __restrict__ lpr_out, float* __restrict__ lmr_out, int keep_taps,
                                                const float* __restrict__ lmr_est_prev, int field_cur, int field_prev) {
    using G = ExtractGeom<TA>;
    constexpr int XS = G::XS, P4 = G::P4, P8 = G::P8, NQ = G::NQ;
    __shared__ __attribute__((aligned(16))) float lpr_ph[4 * P4];     // Re fm_out_iq, 4 phases
    __shared__ __attribute__((aligned(16))) float lmr_re_ph[4 * P4];  // x2-mixed signal, real / imaginary rail
    __shared__ __attribute__((aligned(16))) float lmr_im_ph[4 * P4];
    __shared__ __attribute__((aligned(16))) float rds_re_ph[8 * P8];  // x3-mixed signal
    __shared__ __attribute__((aligned(16))) float rds_im_ph[8 * P8];
    __shared__ __attribute__((aligned(16))) float res_lpr[TA];
    __shared__ __attribute__((aligned(16))) float res_lmr[TA];
    __shared__ __attribute__((aligned(16))) float res_rds[TA];        // [TA/2][2]
    __shared__ float res_est_re[G::NEST];

    const int tiles = d.n_audio / TA;
    const int c = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int i0 = tile * TA;
    const int tid = threadIdx.x;
    const int s_lo = 4 * i0 - 124;               // first fm_out sample staged (block relative)
    const int n = d.n_fm_out;
    const float2* x_c = fm_out_iq + (size_t)c * n;
    const float* dt_c = pll_dt + (size_t)c * n;
    // L-R phase offsets: state field (k & 1) holds P_k, the offset block k was mixed with.  lmr_est_prev != nullptr: this launch
    // derives its own P_b from P_{b-1} and the previous block's estimates (done by the first wavefront
    // while the tile's samples are in flight) and tile 0 publishes it; otherwise k_lmr_phase has written P_b behind block b-1.
    const float off_prev = st(state, field_prev, d.C, c);
    float off_cur = lmr_est_prev ? 0.0f : st(state, field_cur, d.C, c);
    __shared__ float off_s;

    // stage + mix: all loads first, then the arithmetic
    {
        constexpr int PER = (XS + TA - 1) / TA;
        float2 xv[PER]; float dv[PER];
        float ev[kLmrInlineMax / kWave];
        if (lmr_est_prev && tid < kWave) {
#pragma unroll
            for (int k = 0; k < kLmrInlineMax / kWave; k++)
                ev[k] = (tid + kWave * k < d.n_est) ? lmr_est_prev[(size_t)c * d.n_est + tid + kWave * k] : 0.0f;
        }
        if (tile != 0) {
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int e = tid + TA * r;
                if (e < XS) { xv[r] = x_c[s_lo + e]; dv[r] = dt_c[s_lo + e]; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int e = tid + TA * r;
                if (e < XS) {
                    const int s = s_lo + e;
                    xv[r] = (s < 0) ? iq_tail_in[(size_t)c * 128 + 128 + s] : x_c[s];
                    dv[r] = (s < 0) ? dt_tail_in[(size_t)c * 128 + 128 + s] : dt_c[s];
                }
            }
        }
        if (lmr_est_prev) {
            if (tid < kWave) {
                // (tolerance mode: the block's estimates summed as a tree, not in sample order — ~20 instructions in one wavefront
                // of every tile instead of n_est dependent additions)
                float part = 0.0f;
#pragma unroll
                for (int k = 0; k < kLmrInlineMax / kWave; k++) part += ev[k];
                // avg by a reciprocal; |phase + 0.1 avg| < 2 pi + 0.16, so the fmod is at most one exact subtraction
                float nxt = fmaf(wave_sum_f32(part) * __builtin_amdgcn_rcpf((float)d.n_est), 0.1f, off_prev);   // every lane the same value
                const float two_pi = bits_f32(kTwoPiBits);
                nxt = (nxt >= two_pi) ? nxt - two_pi : ((nxt <= -two_pi) ? nxt + two_pi : nxt);
                if (tid == 0) { off_s = nxt; if (tile == 0) st(state, field_cur, d.C, c) = nxt; }
            }
            __syncthreads();
            off_cur = off_s;
        }
        // FMD_FLAG_FAST_MATH: one hardware sine / cosine of the pilot phase per sample, the x2 and x3 harmonics by angle
        // multiplication, the L-R phase offset as a rotation by a per-block constant
        float co_cur = 1.f, so_cur = 0.f, co_prev = 1.f, so_prev = 0.f;
        if constexpr (FAST) {
            co_cur = fast_cos_turns(off_cur); so_cur = fast_sin_turns(off_cur);
            co_prev = fast_cos_turns(off_prev); so_prev = fast_sin_turns(off_prev);
        }
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int e = tid + TA * r;
            if (e < XS) {
                const bool hist = s_lo + e < 0;                          // history samples were mixed with last block's offset
                float2 m2, m3;
                if constexpr (FAST) {
                    const float c1 = fast_cos_turns(dv[r]), s1 = fast_sin_turns(dv[r]);
                    const float c2 = fmaf(c1, c1, -(s1 * s1)), s2 = (c1 + c1) * s1;
                    const float co = hist ? co_prev : co_cur, so = hist ? so_prev : so_cur;
                    const float c2o = fmaf(c2, co, -(s2 * so)), s2o = fmaf(s2, co, c2 * so);
                    m2 = make_float2(fmaf(c2o, xv[r].x, -(xv[r].y * s2o)), fmaf(c2o, xv[r].y, xv[r].x * s2o));
                    const float c3 = fmaf(c2, c1, -(s2 * s1)), s3 = fmaf(s2, c1, c2 * s1);
                    m3 = make_float2(fmaf(c3, xv[r].x, -(xv[r].y * s3)), fmaf(c3, xv[r].y, xv[r].x * s3));
                } else {
                    const float off = hist ? off_prev : off_cur;
                    m2 = harmonic_mix(xv[r], dv[r], 2.0f, off, off + 0.25f);
                    m3 = harmonic_mix(xv[r], dv[r], 3.0f, 0.0f, 0.25f);
                }
                const int a4 = (e & 3) * P4 + (e >> 2);
                lpr_ph[a4] = xv[r].x;
                lmr_re_ph[a4] = m2.x;
                lmr_im_ph[a4] = m2.y;
                if (e >= 4) {
                    const int e8 = e - 4, a8 = (e8 & 7) * P8 + (e8 >> 3);
                    rds_re_ph[a8] = m3.x;
                    rds_im_ph[a8] = m3.y;
                }
            }
        }
    }
    __syncthreads();

    const int job = tid / NQ, u = tid % NQ;
    const float* taps_lpr = b_lpr + (size_t)c * 128;
    const float* taps_lmr = b_lmr + (size_t)c * 128;
    const int est_first = (10 - (i0 % 10)) % 10;   // first output of this tile whose block index is a multiple of 10
    if (job == 0) {
        float o[4];
        fir4_quad(lpr_ph, P4, u, [&](int k) { return taps_lpr[k]; }, o);
        *reinterpret_cast<float4*>(res_lpr + 4 * u) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (job == 1) {
        float o[4];
        fir4_quad(lmr_im_ph, P4, u, [&](int k) { return taps_lmr[k]; }, o);
        *reinterpret_cast<float4*>(res_lmr + 4 * u) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (job == 2) {
        // RDS: first half of the job's threads take the real rail, second half the imaginary rail
        const int rail = u / (NQ / 2), uu = u % (NQ / 2);
        float o[4];
        fir8_quad(rail ? rds_im_ph : rds_re_ph, P8, uu, [&](int k) { return rds_taps.b[k]; }, o);
#pragma unroll
        for (int v = 0; v < 4; v++) res_rds[2 * (4 * uu + v) + rail] = o[v];
    } else {
        // real rail of the L-R outputs that feed the phase estimate (one output per thread)
        const int ii = est_first + 10 * u;
        if (ii < TA) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int jj = 0; jj < 32; jj++) {
#pragma unroll
                for (int p = 0; p < 4; p++) a[p] = fmaf(lmr_re_ph[p * P4 + ii + jj], taps_lmr[4 * jj + p], a[p]);
            }
            res_est_re[u] = (a[0] + a[2]) + (a[1] + a[3]);
        }
    }
    __syncthreads();

    {
        const int i = i0 + tid;
        const float lpr = res_lpr[tid], lmr = res_lmr[tid];
        const int mode = (int)mixctl[2 * c];
        const float kmix = mixctl[2 * c + 1];
        float l, r;
        if (mode == FMD_AUDIO_STEREO) { l = fmaf(lmr, kmix, lpr); r = fmaf(-lmr, kmix, lpr); }
        else if (mode == FMD_AUDIO_LMR) { l = lmr; r = lmr; }
        else { l = lpr; r = lpr; }
        reinterpret_cast<float2*>(audio)[(size_t)c * d.n_audio + i] = make_float2(l + l, r + r);
        if (keep_taps) { lpr_out[(size_t)c * d.n_audio + i] = lpr; lmr_out[(size_t)c * d.n_audio + i] = lmr; }
        if (tid < TA / 2) rds[(size_t)c * d.n_rds + i0 / 2 + tid] = make_float2(res_rds[2 * tid], res_rds[2 * tid + 1]);
        // reference :500-510: estimate against the +-pi/2 constellation, every 10th output of the block
        const int ii = est_first + 10 * tid;
        if (ii < TA) {
            const float ph = FAST ? fast_atan2f(res_lmr[ii], res_est_re[tid]) : fmd_atan2f(res_lmr[ii], res_est_re[tid]);
            const float half_pi = bits_f32(kHalfPiBits);
            lmr_est[(size_t)c * d.n_est + (i0 + ii) / 10] = (ph > 0.0f) ? (half_pi - ph) : (-half_pi - ph);
        }
    }
    // history for the next block: last 128 fm_out_iq / pll_dt samples
    if (tile == tiles - 1 && tid < 128) {
        iq_tail_out[(size_t)c * 128 + tid] = x_c[n - 128 + tid];
        dt_tail_out[(size_t)c * 128 + tid] = dt_c[n - 128 + tid];
    }
}

#ifdef FMD_X_PROBE
// development probe (tools/dbg/bp_probe.py): cycles of k_extract_bp's phases, summed over sampled workgroups (wavefronts 0 and 3), and their count
__device__ unsigned long long g_x_probe[16];
__device__ unsigned long long g_x_probe2[16];
#define X_STAMP(i_) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (xp_on && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 3)) \
    atomicAdd(&g_x_probe[(i_) + ((threadIdx.x >> 6) ? 8 : 0)], t_ - xp_t); xp_t = t_; } while (0)
#else
#define X_STAMP(i_)
#endif

#include "fmd_kernels_bp.inc"
#include "fmd_kernels_chain.inc"

// Tolerance mode, FMD_FLAG_KEEP_TAPS (the "fm_out_iq" getter) and block lengths whose audio blocks are not multiples of 256 (those run
// k_extract<128, true> on the interleaved stream, with the history tails it keeps for itself): the analytic signal from the fm_out plane,
// Hilbert rail as k_hilbert makes it.
__global__ __launch_bounds__(256) void k_planes_to_iq(Dims d, const float* __restrict__ fo_pl, float2* __restrict__ fm_out_iq, FrontTaps taps) {
    const int n = d.n_fm_out, tiles = n / 256, c = blockIdx.x / tiles;     // (n_fm_out is a multiple of 512)
    const int s_ = (blockIdx.x % tiles) * 256 + threadIdx.x;
    const float* x = fo_pl + (size_t)c * (kFoPad + n) + kFoPad + s_ - 64;  // im[s] = sum_k b[k] fm_out[s - 64 + k], k odd
    float l1 = 0.f, l3 = 0.f, l5 = 0.f, l7 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        l1 = fmaf(x[1 + 8 * k], taps.b_hilbert_odd[4 * k + 0], l1);
        l3 = fmaf(x[3 + 8 * k], taps.b_hilbert_odd[4 * k + 1], l3);
        l5 = fmaf(x[5 + 8 * k], taps.b_hilbert_odd[4 * k + 2], l5);
        l7 = fmaf(x[7 + 8 * k], taps.b_hilbert_odd[4 * k + 3], l7);
    }
    fm_out_iq[(size_t)c * n + s_] = make_float2(x[32], (l1 + l5) + (l3 + l7));
}

// a11 — reference ExtractComponents :511-516: integrate the mean L-R phase error of the block (sequential sum in sample order).
// Tolerance mode, blocks of up to 10 * kLmrInlineMax audio samples: the NEXT block's k_extract does it for itself
// (in its prologue — a kernel of its own behind k_extract put two launch gaps per block on the one chain of that
// pipeline that cannot overlap with itself).  Exact mode, longer blocks, and the "lmr_phase" getter: this kernel,
// P_next = update(state[field_in], estimates) into out_row[c] (a state field, or a scratch row for the getter).
__global__ __launch_bounds__(kWave) void k_lmr_phase(Dims d, const float* __restrict__ lmr_est, const float* __restrict__ state, int field_in,
                                                     float* __restrict__ out_row) {
    constexpr int TILE = 128;                            // estimates per channel staged at a time
    __shared__ float est[kWave][TILE + 1];
    const int c0 = blockIdx.x * kWave, c = c0 + threadIdx.x;
    const int rows = (d.C - c0) < kWave ? (d.C - c0) : kWave;
    float sum = 0.0f;
    for (int base = 0; base < d.n_est; base += TILE) {
        const int w = (d.n_est - base) < TILE ? (d.n_est - base) : TILE;
        __syncthreads();
        for (int i = threadIdx.x; i < rows * w; i += kWave) {     // row-contiguous reads
            const int r = i / w, col = i - r * w;
            est[r][col] = lmr_est[(size_t)(c0 + r) * d.n_est + base + col];
        }
        __syncthreads();
        for (int i = 0; i < w; i++) sum = sum + est[threadIdx.x][i];   // sequential, in sample order
    }
    if (c >= d.C) return;
    out_row[c] = lmr_phase_finish(sum, d.n_est, state[(size_t)field_in * d.C + c]);
}

// Tolerance-mode form of k_lmr_phase for the batches whose k_extract does not integrate the L-R phase itself: one wavefront per
// station doing exactly what that prologue does (same partial sums, same butterfly, same finish), so a station's outputs do not
// depend on which side of the batch-size switch it runs.
__global__ __launch_bounds__(kWave) void k_lmr_phase_fast(Dims d, const float* __restrict__ lmr_est, const float* __restrict__ state, int field_in,
                                                          float* __restrict__ out_row) {
    const int c = blockIdx.x, tid = threadIdx.x;
    float part = 0.0f;
#pragma unroll
    for (int k = 0; k < kLmrInlineMax / kWave; k++)
        part += (tid + kWave * k < d.n_est) ? lmr_est[(size_t)c * d.n_est + tid + kWave * k] : 0.0f;
    float nxt = fmaf(wave_sum_f32(part) * __builtin_amdgcn_rcpf((float)d.n_est), 0.1f, state[(size_t)field_in * d.C + c]);
    const float two_pi = bits_f32(kTwoPiBits);
    nxt = (nxt >= two_pi) ? nxt - two_pi : ((nxt <= -two_pi) ? nxt + two_pi : nxt);
    if (tid == 0) out_row[c] = nxt;
}

// =============================================================================================
// k_rds_sync — reference ExtractComponents :511-516 (phase integrate), SynchroniseRDS :538-547,
// AGC_Filter (agc.h:12-30), BPSK_Synchroniser::Process (bpsk_synchroniser.cpp:94-186) with TED_Clock
// (ted_clock.cpp:18-44), Zero_Crossing_Detector, Trigger_Cooldown, and the differential Manchester
// decoder (rds_decoder/differential_manchester_decoder.h:32-60).  Lane per channel.
// =============================================================================================
// Workgroup = two wavefronts for 64 channels (round 2): wave 1 only LOADS — the block's RDS baseband twice over (AGC power pass,
// then the synchroniser pass), 32-sample chunks through a four-slot LDS ring, two chunks ahead — wave 0 runs the serial loops,
// lane per channel, and only STORES (symbols; the post-AGC signal for FMD_FLAG_KEEP_TAPS).  See k_deemphasis for why the two
// kinds of traffic must not share a wave on gfx9.  The differential Manchester decoder no longer runs inside the per-sample
// loop: a symbol fires in some lane on almost every sample, so the wavefront walked the (masked) decoder ~1000 times a block for
// ~150 symbols per lane; the loop now only packs the symbols' signs into a per-lane bit string (LDS), and the decoder runs once
// per symbol behind it.  Same state machine, same bytes.
template <bool FAST>
__global__ __launch_bounds__(2 * kWave) void k_rds_sync(Dims d, float2* __restrict__ rds,
                                                        float* __restrict__ state, LoopCoeffs k, float* __restrict__ rds_sym,
                                                        float2* __restrict__ rds_raw_sym, int* __restrict__ rds_count,
                                                        uint8_t* __restrict__ rds_bytes, int* __restrict__ rds_bytes_count,
                                                        int bytes_cap, int keep_taps, const float* __restrict__ rds_pow, int n_pow, TapPtrs taps) {
    // Tolerance mode: two ring slots instead of four (the loader keeps two chunks in registers beyond the one it stores: 7 us of
    // look-ahead are enough): 43 KB of LDS instead of 78 on the 64 CUs this kernel's workgroups sit on for most of a block's time —
    // the extract stage gets three workgroups beside it there instead of two
    constexpr int kRingSlots = FAST ? 2 : 4;
    constexpr int kSignWords = 33;                                       // 1024 symbols per lane between two runs of the decoder (+1: odd stride, no bank conflicts)
    __shared__ __attribute__((aligned(16))) float2 ring[kRingSlots][kWave * kRowC];
    __shared__ unsigned sign_bits[kWave * kSignWords];
    const bool loader = threadIdx.x >= kWave;                            // wave-uniform
    const int lane = threadIdx.x & (kWave - 1), c0 = blockIdx.x * kWave, c = c0 + lane;
    const bool live = c < d.C;
    const int cs = live ? c : d.C - 1;
    // rds_pow (tolerance mode with k_extract_bp): the block's power arrives as n_pow partial sums per station and the block is
    // walked once; otherwise twice (power pass, then the synchroniser pass)
    const bool one_pass = FAST && rds_pow != nullptr;
    const int n = d.n_rds, chunks = n / kChunk, steps = one_pass ? chunks : 2 * chunks;      // step i handles chunk i mod chunks
    if (loader) {
        auto at = [&](int i) { return (i < steps ? i % chunks : chunks - 1) * kChunk; };
        ChunkRegsC ra = chunk_load_c(rds, n, c0, d.C, at(0));
        ChunkRegsC rb = chunk_load_c(rds, n, c0, d.C, at(1));
        if constexpr (FAST) {
            chunk_store_c(ra, ring[0]);
            ra = chunk_load_c(rds, n, c0, d.C, at(2));
            __syncthreads();                                             // step 0 is in the ring
            for (int i = 0; i < steps; i += 2) {                         // while wave 0 works on step i: step i + 1 into the other slot
                chunk_store_c(rb, ring[(i + 1) & 1]);
                rb = chunk_load_c(rds, n, c0, d.C, at(i + 3));
                __syncthreads();
                chunk_store_c(ra, ring[i & 1]);                          // (step i + 2, while wave 0 works on step i + 1)
                ra = chunk_load_c(rds, n, c0, d.C, at(i + 4));
                __syncthreads();
            }
            return;
        }
        chunk_store_c(ra, ring[0]);
        ra = chunk_load_c(rds, n, c0, d.C, at(2));
        chunk_store_c(rb, ring[1]);
        rb = chunk_load_c(rds, n, c0, d.C, at(3));
        __syncthreads();                                                 // steps 0 and 1 are in the ring
        for (int i = 0; i < steps; i += 2) {                             // while wave 0 works on step i: step i + 2 into its slot
            chunk_store_c(ra, ring[(i + 2) & (kRingSlots - 1)]);
            ra = chunk_load_c(rds, n, c0, d.C, at(i + 4));
            __syncthreads();
            chunk_store_c(rb, ring[(i + 3) & (kRingSlots - 1)]);
            rb = chunk_load_c(rds, n, c0, d.C, at(i + 5));
            __syncthreads();
        }
        return;
    }
    if constexpr (!FAST) __builtin_amdgcn_s_setprio(3);   // (tolerance mode: measured +1.5 % without, now that the FIR kernels beside it no longer fight for VALU slots)
    __syncthreads();
    // a13: AGC power pass (reference AGC_Filter::calculate_average_power, agc.h:21-30)
    float power = 0.0f;
    int step = 0;
    if (one_pass) {
        for (int i = 0; i < n_pow; i++) power += rds_pow[(size_t)cs * n_pow + i];
    } else {
        for (; step < chunks; step++) {
            const float2* buf = ring[step & (kRingSlots - 1)];
#pragma unroll 8
            for (int t = 0; t < kChunk; t++) {
                const float2 x = buf[lane * kRowC + t];
                power = power + fmaf(x.x, x.x, x.y * x.y);
            }
            __syncthreads();
        }
    }
    float gain = st(state, S_AGC_RDS_GAIN, d.C, cs);
    {
        const float target_gain = sqrtf((0.5f / power) * (float)n);
        gain = fmaf(target_gain - gain, 0.2f, gain);
    }

    // a14: BPSK synchroniser state
    float pll_x1 = st(state, S_B_PLL_X1, d.C, cs), pll_y1 = st(state, S_B_PLL_Y1, d.C, cs);
    float pll_int = st(state, S_B_PLL_INT, d.C, cs), pll_err = st(state, S_B_PLL_ERR, d.C, cs), mix_t = st(state, S_B_MIX_T, d.C, cs);
    float zcd_xn = st(state, S_B_ZCD_XN, d.C, cs);
    int cooldown = __float_as_int(st(state, S_B_COOLDOWN, d.C, cs));
    float ted_err = st(state, S_B_TED_ERR, d.C, cs), ted_x1 = st(state, S_B_TED_X1, d.C, cs), ted_y1 = st(state, S_B_TED_Y1, d.C, cs);
    float ted_int = st(state, S_B_TED_INT, d.C, cs), clock = st(state, S_B_CLOCK, d.C, cs);
    float dump_r = st(state, S_B_DUMP_R, d.C, cs), dump_i = st(state, S_B_DUMP_I, d.C, cs);
    int n_sym = 0;
    unsigned sign_word = 0u;
    // differential Manchester (reference rds_decoder/differential_manchester_decoder.h:32-60): every second symbol,
    // bit = sign(cur) xor sign(prev), MSB first, bytes handed on in buffers of 16.
    // state: flags = is_read_bit | prev_bit<<1 | bit_index<<2 | byte_index<<5
    unsigned mflags = __float_as_uint(st(state, S_M_FLAGS, d.C, cs));
    unsigned mbuf[4] = {__float_as_uint(st(state, S_M_BUF0, d.C, cs)), __float_as_uint(st(state, S_M_BUF1, d.C, cs)),
                        __float_as_uint(st(state, S_M_BUF2, d.C, cs)), __float_as_uint(st(state, S_M_BUF3, d.C, cs))};
    int n_bytes = 0;
    int wsym = 0;                                        // symbols in the sign buffer (it holds 32 (kSignWords - 1) per lane)
    // the decoder over the buffered signs, then the buffer starts over: behind the sample loop, and inside it whenever a lane's row
    // is nearly full (blocks of more than ~4800 RDS samples; the symbol clock tops out at 3500 Hz of 16 kHz)
    auto decode_buffered = [&]() __attribute__((always_inline)) {
        sign_bits[lane * kSignWords + (wsym >> 5)] = sign_word;            // the last, partial word
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int max_sym = wsym;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) { const int o = __shfl_xor(max_sym, dd, kWave); max_sym = o > max_sym ? o : max_sym; }
        for (int i = 0; i < max_sym; i++) {
            if (i < wsym) {
                const unsigned cur = (sign_bits[lane * kSignWords + (i >> 5)] >> (i & 31)) & 1u;
                mflags ^= 1u;
                if (mflags & 1u) {
                    const unsigned bit = cur ^ ((mflags >> 1) & 1u);
                    mflags = (mflags & ~2u) | (cur << 1);
                    unsigned bit_index = (mflags >> 2) & 7u, byte_index = (mflags >> 5) & 31u;
                    const unsigned word = byte_index >> 2, shift = (byte_index & 3u) * 8u;
                    if (bit_index == 0u) mbuf[word] &= ~(0xffu << shift);
                    mbuf[word] |= (bit << (7u - bit_index)) << shift;
                    bit_index++;
                    byte_index += bit_index >> 3;
                    bit_index &= 7u;
                    if (byte_index == 16u) {
                        byte_index = 0u;
                        if (live && n_bytes + 16 <= bytes_cap) {
                            uint32_t* o = reinterpret_cast<uint32_t*>(rds_bytes + (size_t)c * bytes_cap + n_bytes);
                            o[0] = mbuf[0]; o[1] = mbuf[1]; o[2] = mbuf[2]; o[3] = mbuf[3];
                        }
                        n_bytes += 16;
                    }
                    mflags = (mflags & 3u) | (bit_index << 2) | (byte_index << 5);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        wsym = 0; sign_word = 0u;
    };

    const float Ts = 1.0f / 16e3f;
    const float KTs_pi = (10.0f * Ts) * (2e3f / 16e3f);
    const float half_pi = bits_f32(kHalfPiBits), two_over_pi = bits_f32(kTwoOverPiBits);

    // Tolerance mode: a symbol's phase estimate (arctangent, stores, sign packing: half of the loop's instructions) does not run inside
    // the sample in which the clock wrapped — with 64 stations per wavefront some lane wraps on almost every sample, so every sample
    // paid for it — but once per group of four samples: the symbol clock tops out at 3500 Hz of 16 kHz, so a station has at most
    // one symbol pending per group.  The carrier loop then sees its new phase error up to three samples (0.2 ms) later than the
    // reference's would: at the end of the group the loop filter, integrator and NCO phase are corrected for that — between two
    // symbols the carrier loop is linear in its phase error, so what the j samples behind the wrap would have added under the new
    // error is a constant times the change (dl / di / dm below; the clamps do not act on a loop near lock) — and acquisition runs as
    // it does when every symbol is handled in its own sample.
    if constexpr (FAST) {
        bool pend = false;
        float psr = 0.0f, psi = 0.0f;
        int jw = 0;                                          // samples of the group behind the one in which the clock wrapped
        const float Ts10 = 10.0f * Ts;
        // response of (loop filter output, integrator, NCO phase) after j = 1, 2, 3 samples to a unit step of the phase error
        const float dl1 = k.bpsk_b1, dl2 = fmaf(k.bpsk_a0, dl1, k.bpsk_b0 + k.bpsk_b1), dl3 = fmaf(k.bpsk_a0, dl2, k.bpsk_b0 + k.bpsk_b1);
        const float dm1 = Ts10 * fmaf(0.3f, dl1, KTs_pi), dm2 = dm1 + Ts10 * fmaf(0.3f, dl2, 2.0f * KTs_pi), dm3 = dm2 + Ts10 * fmaf(0.3f, dl3, 3.0f * KTs_pi);
        for (; step < steps; step++) {
            const int ch = one_pass ? step : step - chunks;
            float2* buf = ring[step & (kRingSlots - 1)];
            for (int t4 = 0; t4 < kChunk; t4 += 4) {
                // (this wavefront is alone on its SIMD and issue-bound at ~5 cycles per instruction, whatever the instruction: what counts
                //  is their number.  Within a group the phase error is constant: the loop filter's input term is computed once per group —
                //  the first sample may still see the previous error in x1 —, the integrator steps by a constant, and the NCO phase is
                //  wrapped once per group: v_sin / v_cos take arguments up to 256 turns, the phase moves < 1e-3 turns per sample.)
                const float lpf_in = pll_err * (k.bpsk_b0 + k.bpsk_b1), lpf_in0 = fmaf(pll_x1, k.bpsk_b0, pll_err * k.bpsk_b1), int_step = pll_err * KTs_pi;
                pll_x1 = pll_err;
                mix_t = mix_t - rintf(mix_t);
#pragma unroll
                for (int u4 = 0; u4 < 4; u4++) {
                    const int t = t4 + u4;
                    const float2 xr = buf[lane * kRowC + t];
                    const float p = gain * xr.x, q = gain * xr.y;
                    if (keep_taps) buf[lane * kRowC + t] = make_float2(p, q);
                    // carrier PLL PI controller; PLL_Mixer::Update: f_center 0, f_gain 10
                    const float pll_lpf = fmaf(pll_y1, k.bpsk_a0, u4 ? lpf_in : lpf_in0);
                    pll_y1 = pll_lpf;
                    pll_int = clampf(pll_int + int_step, -1.0f, 1.0f);
                    const float control = clampf(fmaf(pll_lpf, 0.3f, pll_int), -1.0f, 1.0f);
                    mix_t = fmaf(control, Ts10, mix_t);
                    const float ps = fast_sin_turns(mix_t), pc = fast_cos_turns(mix_t);
                    const float iq_r = fmaf(pc, p, -(q * ps));
                    const float iq_i = fmaf(p, ps, q * pc);
                    // zero crossing on Q with hold-off
                    bool is_zcd = 0.0f > (iq_i * zcd_xn);
                    zcd_xn = iq_i;
                    if (is_zcd && cooldown == 0) { cooldown = 4; }
                    else { if (cooldown > 0) cooldown--; is_zcd = false; }
                    if (is_zcd) { float e2 = clock + clock; if (clock > 0.5f) e2 = e2 - 2.0f; ted_err = e2; }
                    // TED PI controller
                    const float ted_lpf = fmaf(ted_x1, k.ted_b0, fmaf(ted_y1, k.ted_a0, ted_err * k.ted_b1));
                    ted_x1 = ted_err; ted_y1 = ted_lpf;
                    ted_int = clampf(fmaf(ted_err, KTs_pi, ted_int), -1.0f, 1.0f);
                    const float PI_ted = fmaf(ted_lpf, 0.3f, ted_int);
                    // integrate and dump
                    dump_r = fmaf(0.25f, iq_r, dump_r);
                    dump_i = fmaf(0.25f, iq_i, dump_i);
                    // TED_Clock::update: fcenter 2000, fgain 1500
                    const float dd = fmaf(clampf(-PI_ted, -1.0f, 1.0f), 1.5e3f * Ts, 2e3f * Ts);     // (cfreq Ts in one step)
                    const float cy = dd + clock;
                    const bool wrapped = !(fmaf(-dd, 0.5f, 1.0f) > cy);
                    clock = wrapped ? 0.0f : cy;
                    psr = wrapped ? dump_r : psr; psi = wrapped ? dump_i : psi;
                    dump_r = wrapped ? 0.0f : dump_r; dump_i = wrapped ? 0.0f : dump_i;
                    jw = wrapped ? 3 - u4 : jw;
                    pend = pend || wrapped;
                }
                if (pend) {
                    const float phs = fast_atan2f(psi, psr);
                    const float est = (phs > 0.0f) ? (half_pi - phs) : (-half_pi - phs);
                    const float err_new = est * two_over_pi, de = err_new - pll_err;
                    pll_err = err_new;
                    // the jw samples behind the wrap ran on the old error
                    pll_x1 = (jw >= 1) ? err_new : pll_x1;
                    pll_y1 = fmaf((jw == 1) ? dl1 : ((jw == 2) ? dl2 : ((jw == 3) ? dl3 : 0.0f)), de, pll_y1);
                    pll_int = fmaf(KTs_pi * (float)jw, de, pll_int);
                    mix_t = fmaf((jw == 1) ? dm1 : ((jw == 2) ? dm2 : ((jw == 3) ? dm3 : 0.0f)), de, mix_t);
                    if (live) {
                        rds_sym[(size_t)c * n + n_sym] = psi;
                        if (keep_taps) rds_raw_sym[(size_t)c * n + n_sym] = make_float2(psr, psi);
                    }
                    sign_word |= ((psi > 0.0f) ? 1u : 0u) << (wsym & 31);
                    if ((wsym & 31) == 31) { sign_bits[lane * kSignWords + (wsym >> 5)] = sign_word; sign_word = 0u; }
                    n_sym++; wsym++;
                    pend = false;
                }
            }
            if (__builtin_amdgcn_ballot_w64(wsym > 32 * (kSignWords - 1) - kChunk) != 0ull) decode_buffered();
            if (keep_taps) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                chunk_flush_c(buf, rds, n, c0, d.C, ch * kChunk);
            }
            __syncthreads();
        }
    } else {
        for (; step < steps; step++) {
            const int ch = one_pass ? step : step - chunks;
            float2* buf = ring[step & (kRingSlots - 1)];
            for (int t = 0; t < kChunk; t++) {
                const float2 xr = buf[lane * kRowC + t];
                const float p = gain * xr.x, q = gain * xr.y;
                if (keep_taps) buf[lane * kRowC + t] = make_float2(p, q);
                // carrier PLL PI controller
                const float lt0 = fmaf(pll_x1, k.bpsk_b0, pll_y1 * k.bpsk_a0);
                const float pll_lpf = (0.0f + lt0) + fmaf(pll_err, k.bpsk_b1, 0.0f);
                pll_x1 = pll_err; pll_y1 = pll_lpf;
                pll_int = clampf(fmaf(pll_err, KTs_pi, pll_int), -1.0f, 1.0f);
                const float PI_pll = fmaf(pll_lpf, 0.3f, pll_int);
                // PLL_Mixer::Update: f_center 0, f_gain 10
                const float control = clampf(PI_pll * 1.0f, -1.0f, 1.0f);
                const float freq = fmaf(control, 10.0f, 0.0f);
                const float yy = fmaf(freq, Ts, mix_t);
                float ps, pc;
                if constexpr (FAST) {
                    mix_t = yy - rintf(yy);
                    ps = fast_sin_turns(mix_t); pc = fast_cos_turns(mix_t);
                } else {
                    mix_t = yy - round_half_away(yy);
                    float dt_cos = mix_t + 0.25f;
                    dt_cos = dt_cos - round_half_away(dt_cos);
                    ps = cheb_sine_scalar(mix_t); pc = cheb_sine_scalar(dt_cos);
                }
                const float iq_r = fmaf(pc, p, -(q * ps));
                const float iq_i = fmaf(p, ps, q * pc);
                // zero crossing on Q with hold-off
                bool is_zcd = 0.0f > (iq_i * zcd_xn);
                zcd_xn = iq_i;
                if (is_zcd && cooldown == 0) { cooldown = 4; }
                else { if (cooldown > 0) cooldown--; is_zcd = false; }
                if (is_zcd) { float e2 = clock + clock; if (clock > 0.5f) e2 = e2 - 2.0f; ted_err = e2; }
                // TED PI controller
                const float tt0 = fmaf(ted_x1, k.ted_b0, ted_y1 * k.ted_a0);
                const float ted_lpf = (0.0f + tt0) + fmaf(ted_err, k.ted_b1, 0.0f);
                ted_x1 = ted_err; ted_y1 = ted_lpf;
                ted_int = clampf(fmaf(ted_err, KTs_pi, ted_int), -1.0f, 1.0f);
                const float PI_ted = fmaf(ted_lpf, 0.3f, ted_int);
                // integrate and dump
                dump_r = fmaf(0.25f, iq_r, dump_r);
                dump_i = fmaf(0.25f, iq_i, dump_i);
                // TED_Clock::update: fcenter 2000, fgain 1500
                const float ccontrol = clampf((-PI_ted) * 1.0f, -1.0f, 1.0f);
                const float cfreq = fmaf(ccontrol, 1.5e3f, 2e3f);
                const float dd = cfreq * Ts;
                const float cy = dd + clock;
                const float thr = fmaf(-dd, 0.5f, 1.0f);
                const bool is_ted = !(thr > cy);
                if (!is_ted) {
                    clock = cy;
                } else {
                    clock = 0.0f;
                    const float sr = dump_r, si = dump_i;
                    dump_r = 0.0f; dump_i = 0.0f;
                    const float phs = FAST ? fast_atan2f(si, sr) : fmd_atan2f(si, sr);
                    const float est = (phs > 0.0f) ? (half_pi - phs) : (-half_pi - phs);
                    pll_err = est * two_over_pi;
                    if (live) {
                        rds_sym[(size_t)c * n + n_sym] = si;
                        if (keep_taps) rds_raw_sym[(size_t)c * n + n_sym] = make_float2(sr, si);
                    }
                    // the symbol's sign for the Manchester decoder behind the loop
                    sign_word |= ((si > 0.0f) ? 1u : 0u) << (wsym & 31);
                    if ((wsym & 31) == 31) { sign_bits[lane * kSignWords + (wsym >> 5)] = sign_word; sign_word = 0u; }
                    n_sym++; wsym++;
                }
                if constexpr (!FAST) {
                    // FMD_FLAG_KEEP_TAPS: BPSK_Synchroniser's per-sample views (bpsk_synchroniser.h:78-85; the oracle's bpsk_process keeps the same
                    // values at the same point of the iteration)
                    if (taps.b_pll_sym && live) {
                        const size_t o = (size_t)c * n + (size_t)ch * kChunk + t;
                        taps.b_pll_sym[o] = make_float2(iq_r, iq_i); taps.b_intdump[o] = make_float2(dump_r, dump_i);
                        taps.b_ted_raw[o] = ted_err; taps.b_ted_pi[o] = PI_ted; taps.b_pll_raw[o] = pll_err; taps.b_pll_pi[o] = PI_pll;
                        taps.b_zcd[o] = is_zcd ? 1.0f : 0.0f; taps.b_trig[o] = is_ted ? 1.0f : 0.0f;
                    }
                }
            }
            // a chunk adds at most kChunk / 4 symbols to a lane's row (3500 Hz of 16 kHz is fewer): decode before any row can overflow
            if (__builtin_amdgcn_ballot_w64(wsym > 32 * (kSignWords - 1) - kChunk) != 0ull) decode_buffered();
            if (keep_taps) {
                // write the post-AGC RDS signal back (reference GetRDSOutput)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                chunk_flush_c(buf, rds, n, c0, d.C, ch * kChunk);
            }
            __syncthreads();
        }
    }
    decode_buffered();
    if (live) {
        st(state, S_AGC_RDS_GAIN, d.C, c) = gain;
        st(state, S_B_PLL_X1, d.C, c) = pll_x1; st(state, S_B_PLL_Y1, d.C, c) = pll_y1; st(state, S_B_PLL_INT, d.C, c) = pll_int;
        st(state, S_B_PLL_ERR, d.C, c) = pll_err; st(state, S_B_MIX_T, d.C, c) = mix_t; st(state, S_B_ZCD_XN, d.C, c) = zcd_xn;
        st(state, S_B_COOLDOWN, d.C, c) = __int_as_float(cooldown);
        st(state, S_B_TED_ERR, d.C, c) = ted_err; st(state, S_B_TED_X1, d.C, c) = ted_x1; st(state, S_B_TED_Y1, d.C, c) = ted_y1;
        st(state, S_B_TED_INT, d.C, c) = ted_int; st(state, S_B_CLOCK, d.C, c) = clock;
        st(state, S_B_DUMP_R, d.C, c) = dump_r; st(state, S_B_DUMP_I, d.C, c) = dump_i;
        st(state, S_M_FLAGS, d.C, c) = __uint_as_float(mflags);
        st(state, S_M_BUF0, d.C, c) = __uint_as_float(mbuf[0]); st(state, S_M_BUF1, d.C, c) = __uint_as_float(mbuf[1]);
        st(state, S_M_BUF2, d.C, c) = __uint_as_float(mbuf[2]); st(state, S_M_BUF3, d.C, c) = __uint_as_float(mbuf[3]);
        rds_count[c] = n_sym;
        rds_bytes_count[c] = n_bytes;
    }
}

// =============================================================================================
// optional de-emphasis path — reference Run_FM_Demodulate :403-410: IIR_Filter<float> K=2 in place on
// fm_out (serial per channel), then the Hilbert FIR.  k_front writes fm_out when any channel asks for it.
// =============================================================================================
// Lane per channel, 32-sample chunks moved row-wise (coalesced, 16 B per lane) and transposed through LDS, the next chunk in
// flight while the current one is filtered — the form of the other serial kernels.  (Round 1 walked each channel's row straight
// in global memory: 64 lanes, 64 different rows per load.)  The recurrence is 4 dependent operations per sample.
struct ChunkRegsF { float4 v0, v1, v2, v3, v4, v5, v6, v7; };
#define FMD_FOR8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
__device__ __forceinline__ ChunkRegsF chunk_load_f(const float* __restrict__ base, int n, int c0, int C, int t0) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 3, col = lane & 7;
    ChunkRegsF r;
#define FMD_LDF(k) { int ch = c0 + 8 * k + row; ch = ch < C ? ch : C - 1; r.v##k = *reinterpret_cast<const float4*>(base + (size_t)ch * n + t0 + 4 * col); }
    FMD_FOR8(FMD_LDF)
#undef FMD_LDF
    return r;
}
__device__ __forceinline__ void chunk_store_f(const ChunkRegsF& r, float* lds) {
    const int lane = threadIdx.x & (kWave - 1), row = lane >> 3, col = lane & 7;
#define FMD_STF(k) *reinterpret_cast<float4*>(lds + (8 * k + row) * kRowF + 4 * col) = r.v##k;
    FMD_FOR8(FMD_STF)
#undef FMD_STF
}

// Workgroup = two wavefronts for 64 channels: wave 1 only LOADS (HBM -> registers -> LDS ring, two chunks ahead), wave 0 filters
// (lane per channel) and only STORES.  On gfx9 loads and stores share one in-order counter and complete out of order with
// respect to each other, so a wave that has both in flight waits for "everything": with the chunk prefetch and the result
// stores in one wave this kernel paid an HBM round trip per 32-sample chunk (4100 cycles per chunk for 770 cycles of work).
__global__ __launch_bounds__(2 * kWave) void k_deemphasis(Dims d, float* __restrict__ fm_out, const float* __restrict__ deemph, float* __restrict__ state, int stride) {
    constexpr int kRingSlots = 4;
    __shared__ __attribute__((aligned(16))) float ring[kRingSlots][kWave * kRowF];
    const bool loader = threadIdx.x >= kWave;              // wave-uniform
    const int lane = threadIdx.x & (kWave - 1), c0 = blockIdx.x * kWave, c = c0 + lane;
    const bool live = c < d.C;
    const int cs = live ? c : d.C - 1;
    const bool on = deemph[4 * cs + 3] != 0.0f;
    const int n = d.n_fm_out, chunks = n / kChunk;
    // a workgroup none of whose channels filters has nothing to do (both waves see the same 64 channels)
    if (__builtin_amdgcn_ballot_w64(on && live) == 0ull) return;
    if (loader) {
        ChunkRegsF ra = chunk_load_f(fm_out, stride, c0, d.C, 0);
        ChunkRegsF rb = chunk_load_f(fm_out, stride, c0, d.C, kChunk);
        chunk_store_f(ra, ring[0]);
        ra = chunk_load_f(fm_out, stride, c0, d.C, (2 < chunks ? 2 : chunks - 1) * kChunk);
        chunk_store_f(rb, ring[1]);
        rb = chunk_load_f(fm_out, stride, c0, d.C, (3 < chunks ? 3 : chunks - 1) * kChunk);
        __syncthreads();                                   // chunks 0 and 1 are in the ring
        for (int ch = 0; ch < chunks; ch += 2) {           // while wave 0 filters chunk ch: chunk ch + 2 into its slot
            chunk_store_f(ra, ring[(ch + 2) & (kRingSlots - 1)]);
            ra = chunk_load_f(fm_out, stride, c0, d.C, (ch + 4 < chunks ? ch + 4 : chunks - 1) * kChunk);
            __syncthreads();
            chunk_store_f(rb, ring[(ch + 3) & (kRingSlots - 1)]);
            rb = chunk_load_f(fm_out, stride, c0, d.C, (ch + 5 < chunks ? ch + 5 : chunks - 1) * kChunk);
            __syncthreads();
        }
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    const float b0 = deemph[4 * cs + 0], b1 = deemph[4 * cs + 1], a0 = deemph[4 * cs + 2];
    float x1 = st(state, S_DE_X1, d.C, cs), y1 = st(state, S_DE_Y1, d.C, cs);
    __syncthreads();
    for (int ch = 0; ch < chunks; ch++) {
        float* buf = ring[ch & (kRingSlots - 1)];
        float4* row_ = reinterpret_cast<float4*>(buf + lane * kRowF);
        float4 w_[kChunk / 4];
#pragma unroll
        for (int t = 0; t < kChunk / 4; t++) w_[t] = row_[t];
        if (on) {
#pragma unroll
            for (int t = 0; t < kChunk / 4; t++) {
                float xs_[4] = {w_[t].x, w_[t].y, w_[t].z, w_[t].w}, ys_[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    // reference IIR_Filter<float> K=2 (iir_filter.h:62-69): t_i = fma(xn[i], b[i], yn[i] a[i]); y = (0 + t_0) + t_1
                    const float t0 = fmaf(x1, b0, y1 * a0);
                    const float y = (0.0f + t0) + fmaf(xs_[u], b1, 0.0f);
                    x1 = xs_[u]; y1 = y; ys_[u] = y;
                }
                w_[t] = make_float4(ys_[0], ys_[1], ys_[2], ys_[3]);
            }
#pragma unroll
            for (int t = 0; t < kChunk / 4; t++) row_[t] = w_[t];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        chunk_flush_f(buf, fm_out, stride, c0, d.C, ch * kChunk);     // stores only: nothing in this wave ever waits for them
        __syncthreads();
    }
    if (live && on) { st(state, S_DE_X1, d.C, c) = x1; st(state, S_DE_Y1, d.C, c) = y1; }
}

__global__ __launch_bounds__(256) void k_hilbert(Dims d, const float* __restrict__ fm_out, const float* __restrict__ fo_tail_in,
                                                 float* __restrict__ fo_tail_out, float2* __restrict__ fm_out_iq, FrontTaps taps) {
    constexpr int T = 256;
    __shared__ float fo[T + 64];
    const int tiles = d.n_fm_out / T;
    const int c = blockIdx.x / tiles, tile = blockIdx.x % tiles, o0 = tile * T, tid = threadIdx.x;
    const float* row = fm_out + (size_t)c * d.n_fm_out;
    for (int uu = tid; uu < T + 64; uu += 256) {
        const int u = o0 - 64 + uu;
        fo[uu] = (u < 0) ? fo_tail_in[(size_t)c * 64 + 64 + u] : row[u];
    }
    __syncthreads();
    const int oo = tid;
    float l1 = 0.f, l3 = 0.f, l5 = 0.f, l7 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        l1 = fmaf(fo[oo + 1 + 8 * k], taps.b_hilbert_odd[4 * k + 0], l1);
        l3 = fmaf(fo[oo + 3 + 8 * k], taps.b_hilbert_odd[4 * k + 1], l3);
        l5 = fmaf(fo[oo + 5 + 8 * k], taps.b_hilbert_odd[4 * k + 2], l5);
        l7 = fmaf(fo[oo + 7 + 8 * k], taps.b_hilbert_odd[4 * k + 3], l7);
    }
    const float im = (0.0f + ((l1 + l5) + (l3 + l7))) + 0.0f;
    fm_out_iq[(size_t)c * d.n_fm_out + o0 + oo] = make_float2(fo[oo + 32], im);
    if (tile == tiles - 1 && tid < 64) fo_tail_out[(size_t)c * 64 + tid] = row[d.n_fm_out - 64 + tid];
}

#include "fmd_kernels_fast.inc"

__global__ void k_selftest_atan2(const float* __restrict__ y, const float* __restrict__ x, float* __restrict__ out,
                                 unsigned char* __restrict__ ok_out, size_t n, int table_form) {
    __shared__ AtanTable atab;
    atan_table_fill(&atab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (table_form == 10) { out[i] = fast_atan2f(y[i], x[i]); }                  // FMD_FLAG_FAST_MATH primitives
    else if (table_form == 13) { out[i] = fast_atan2_turns(y[i], x[i]); }
    else if (table_form == 11) { out[i] = fast_sin_turns(y[i]); }
    else if (table_form == 12) { out[i] = fast_cos_turns(y[i]); }
    else if (table_form == 2) {   // k_front's discriminator on u8 IQ (operands are small integers)
        out[i] = fmd_atan2f_table<true>(y[i], x[i], &atab);
    } else if (table_form) {   // k_front's discriminator
        out[i] = fmd_atan2f_table(y[i], x[i], &atab);
    } else if (ok_out) {   // the locked-loop short form of k_pilot_pll's phase detector and its domain predicate
        LoopCoeffs k{};
        const PllConsts c = make_pll_consts(k);
        bool ok;
        out[i] = atan2f_locked(y[i], x[i], c, ok);
        ok_out[i] = ok ? 1 : 0;
    } else {
        out[i] = fmd_atan2f(y[i], x[i]);
    }
}

hipError_t selftest_atan2(const float* d_y, const float* d_x, float* d_out, unsigned char* d_ok, size_t n, int table_form, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_atan2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_y, d_x, d_out, d_ok, n, table_form);
    return hipGetLastError();
}

// The audio block as the 16-bit PCM frames the reference's scraper writes (fm_scraper.cpp:79-82 through Frame<float> ->
// Frame<int16_t>): sample * (32767 * 0.95f), truncated toward zero.  Half the bytes of the f32 block: what the multi-GPU
// audio gather moves.
__global__ __launch_bounds__(256) void k_audio_pcm16(const float4* __restrict__ in, short4* __restrict__ out, size_t n4) {
    const float scale = 32767.0f * 0.95f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        out[i] = make_short4((short)(int)(v.x * scale), (short)(int)(v.y * scale), (short)(int)(v.z * scale), (short)(int)(v.w * scale));
    }
}

hipError_t launch_audio_pcm16(const float* d_audio, int16_t* d_pcm, size_t n_values, hipStream_t s) {
    const size_t n4 = n_values / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_audio_pcm16, dim3(blocks ? blocks : 1), dim3(256), 0, s, reinterpret_cast<const float4*>(d_audio), reinterpret_cast<short4*>(d_pcm), n4);
    return hipGetLastError();
}

// fresh-construction state (reference constructors: AGC gain 0.1 agc.h:10, everything else zero)
__global__ void k_reset(Dims d, float* __restrict__ state) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.C) return;
    for (int f = 0; f < S_NUM_FIELDS; f++) st(state, f, d.C, c) = 0.0f;
    st(state, S_AGC_PILOT_GAIN, d.C, c) = 0.1f;
    st(state, S_AGC_RDS_GAIN, d.C, c) = 0.1f;
}

// ---------------------------------------------------------------------------------------------
// host-side stage launchers
// ---------------------------------------------------------------------------------------------
// Batches of 1024 .. 1792 stations at 256 kSa/s (the smallest on the deferred schedule — fmd_api.cpp — up to where the front end's own length takes over): the serial RDS stage's launch is the step (0.092 ms alone, 0.12 beside the throughput kernels)
// and the front end has slack.  With 44 KB of extra dynamic LDS its workgroups come two to a CU instead of six and leave the serial stages'
// wavefronts the issue slots: 1024 stations 0.125 -> 0.114 ms (RDS launch 0.121 -> 0.102, the front end itself 0.060 -> 0.053), 1536 stations
// 0.135 -> 0.126, 1792: 0.144 -> 0.136, 1024 x u8 0.117 -> 0.110; 512 stations and the 40 of configs[4]: no difference; 640: slower (0.111 ->
// 0.120: no deferred schedule there), 2048: slower (0.148 -> 0.154) (profiles/round5/rds_stage_ab.txt).  FMD_FRONT_LDS_PAD (development builds) overrides the amount.
static size_t front_lds_pad(const Dims& d) {
    static const long forced = dev_env("FMD_FRONT_LDS_PAD") ? atol(dev_env("FMD_FRONT_LDS_PAD")) : -1;
    if (forced >= 0) return (size_t)(forced <= 49152 ? forced : 49152);
    return (d.m == 1 && d.C >= 1024 && d.C <= 1792) ? 45056 : 0;
}
template <typename InT, int TT = 512, bool FAST = false>
static hipError_t launch_front(const LaunchCtx& ctx, SlotRef r, const InT* d_iq, hipStream_t s, const SlotRef* pll = nullptr) {
    using G = FrontGeom<TT>;
    const Dims& d = ctx.d;
    if constexpr (!FAST) {
        if (ctx.fast) return launch_front<InT, TT, true>(ctx, r, d_iq, s, pll);
    }
    if constexpr (TT == 512) {
        if constexpr (FAST && sizeof(InT) == 8) {
            // cf32 captures, round 6: 2048-output tiles for batches that still give the chip 12 288 workgroups (3072 stations' worth of 64 ms blocks) —
            // half as many workgroups fetch the FIR's operand images (7 KB each from L2: 230 MB a block at 1024-output tiles) and recompute a
            // tile's 63-sample halo: k_front_mfma 0.154 -> 0.142 ms in the step, 268 -> 278 GSa/s at 4096 stations, +2.5 % at 8192; at 2048 stations
            // (8192 workgroups) 1024-output tiles are 1 % better (profiles/round6/front_tile_ab.txt).  FMD_FRONT_CF32_TILE (development builds) forces one.
            static const int forced = dev_env("FMD_FRONT_CF32_TILE") ? atoi(dev_env("FMD_FRONT_CF32_TILE")) : 0;
            const bool big = forced ? forced == 2048 : (long)d.C * (d.n_fm_out / 2048) >= 12288;
            if (big && d.n_fm_out % 2048 == 0 && !ctx.deemph_in_tile) return launch_front<InT, 2048, FAST>(ctx, r, d_iq, s, pll);
        }
        if constexpr (FAST && sizeof(InT) == 2) {   // u8 captures: 2048-output tiles (4 KB of input per 1024-output workgroup leaves too few bytes in flight per CU)
            static const bool t1024 = dev_env("FMD_FRONT_U8_T1024") != nullptr;      // (A/B hook)
            if (d.n_fm_out % 2048 == 0 && !ctx.deemph_in_tile && !t1024) return launch_front<InT, 2048, FAST>(ctx, r, d_iq, s, pll);
        }
        if (d.n_fm_out % 1024 == 0) return launch_front<InT, 1024, FAST>(ctx, r, d_iq, s, pll);
    }
    const int tiles = d.n_fm_out / G::T;
    if constexpr (FAST) {   // tolerance mode: k_front_mfma; with the de-emphasis IIR inside the tile when a channel asks for it
        PllFusedArgs pf{};
        if (pll) {          // the pilot stage of the block in slot pll->buf as this launch's first workgroups
            const int nxt = (pll->buf + 1) % kSlots;
            pf = PllFusedArgs{(d.C + 31) / 32, ctx.b.pv_pl[pll->buf], ctx.b.pv_hist[pll->par], ctx.b.pv_hist[pll->par ^ 1], ctx.b.fo_pl[pll->buf], ctx.b.fo_pl[nxt],
                              ctx.b.pll_poly[pll->buf], ctx.b.pll_poly[nxt], ctx.b.state, ctx.loops, ctx.b.sparse_tab, ctx.b.spec_stats};
        }
        const unsigned grid = (unsigned)(tiles * d.C + pf.n_wg);
        if (ctx.deemph_in_tile) {
            using GM = FrontGeomM<TT, kDeemphWarmup>;
            if (pll) FMD_LAUNCH(r, true, true, (k_front_mfma<InT, TT, kDeemphWarmup, true>), dim3(grid), dim3(256), sizeof(float) * GM::LDS_FLOATS + front_lds_pad(d), s, d, d_iq,
                                ctx.b.base_tail[r.par], ctx.b.base_tail[r.par ^ 1], ctx.b.fo_pl[r.buf], ctx.front.fm_gain, ctx.b.deemph, ctx.b.front_mfma, ctx.b.pv_pl[r.buf], ctx.b.sparse_tab, pf);
            else FMD_LAUNCH(r, true, true, (k_front_mfma<InT, TT, kDeemphWarmup, false>), dim3(grid), dim3(256), sizeof(float) * GM::LDS_FLOATS + front_lds_pad(d), s, d, d_iq,
                            ctx.b.base_tail[r.par], ctx.b.base_tail[r.par ^ 1], ctx.b.fo_pl[r.buf], ctx.front.fm_gain, ctx.b.deemph, ctx.b.front_mfma, ctx.b.pv_pl[r.buf], ctx.b.sparse_tab, pf);
        } else {
            using GM = FrontGeomM<TT, 0>;
            if (pll) FMD_LAUNCH(r, true, true, (k_front_mfma<InT, TT, 0, true>), dim3(grid), dim3(256), sizeof(float) * GM::LDS_FLOATS + front_lds_pad(d), s, d, d_iq,
                                ctx.b.base_tail[r.par], ctx.b.base_tail[r.par ^ 1], ctx.b.fo_pl[r.buf], ctx.front.fm_gain, ctx.b.deemph, ctx.b.front_mfma, ctx.b.pv_pl[r.buf], ctx.b.sparse_tab, pf);
            else FMD_LAUNCH(r, true, true, (k_front_mfma<InT, TT, 0, false>), dim3(grid), dim3(256), sizeof(float) * GM::LDS_FLOATS + front_lds_pad(d), s, d, d_iq,
                            ctx.b.base_tail[r.par], ctx.b.base_tail[r.par ^ 1], ctx.b.fo_pl[r.buf], ctx.front.fm_gain, ctx.b.deemph, ctx.b.front_mfma, ctx.b.pv_pl[r.buf], ctx.b.sparse_tab, pf);
        }
        return hipGetLastError();
    }
    const size_t lds = sizeof(float) * G::LDS_FLOATS;
    auto kern = k_front<InT, TT>;
    FMD_LAUNCH(r, true, true, kern, dim3((unsigned)(tiles * d.C)), dim3(256), lds, s, d, d_iq, ctx.b.base_tail[r.par], ctx.b.base_tail[r.par ^ 1],
                       ctx.b.fm_out_iq[r.buf], ctx.b.fm_out[r.buf], ctx.b.fo_tail[r.par ^ 1], ctx.front, ctx.any_deemph);
    return hipGetLastError();
}

template <int M, typename InT>
static hipError_t launch_predecim(const LaunchCtx& ctx, SlotRef r, const InT* d_iq, hipStream_t s) {
    const Dims& d = ctx.d;
    using G = PredecimGeom<M>;
    static const bool valu_form = dev_env("FMD_PREDECIM_VALU") != nullptr;      // (A/B hook: the tolerance mode's first decimator on the VALU)
    // tile: 2048 input samples of cf32 (16 KB; the kernel then sits on its HBM floor whatever the tile), 8192 of u8 (16 KB as well:
    // with 4 KB per workgroup too few bytes are in flight per CU); blocks that are no multiple of it: 256 outputs per workgroup
    constexpr int TPM = (sizeof(InT) == 2 ? 8192 : 2048) / M;
    if (ctx.fast && !valu_form && d.n_fm_in % TPM != 0 && d.n_fm_in % 256 == 0) {
        FMD_LAUNCH(r, true, true, (k_predecim_mfma<M, InT, 256>), dim3((unsigned)(d.n_fm_in / 256 * d.C)), dim3(256), 0, s, d, d_iq, ctx.b.pre_tail[r.par], ctx.b.pre_tail[r.par ^ 1],
                   reinterpret_cast<float*>(ctx.b.fm_in[r.buf]), ctx.b.front_mfma + kFrontImgU4);
    } else if (ctx.fast && !valu_form && d.n_fm_in % TPM == 0) {
        FMD_LAUNCH(r, true, true, (k_predecim_mfma<M, InT, TPM>), dim3((unsigned)(d.n_fm_in / TPM * d.C)), dim3(256), 0, s, d, d_iq, ctx.b.pre_tail[r.par], ctx.b.pre_tail[r.par ^ 1],
                   reinterpret_cast<float*>(ctx.b.fm_in[r.buf]), ctx.b.front_mfma + kFrontImgU4);
    } else if (ctx.fast)
        FMD_LAUNCH(r, true, true, (k_predecim<M, InT, true>), dim3((unsigned)(d.n_fm_in / G::TP * d.C)), dim3(256), 0, s, d, d_iq, ctx.b.pre_tail[r.par], ctx.b.pre_tail[r.par ^ 1],
                   ctx.b.fm_in[r.buf], ctx.front);
    else
        FMD_LAUNCH(r, true, true, (k_predecim<M, InT, false>), dim3((unsigned)(d.n_fm_in / G::TP * d.C)), dim3(256), 0, s, d, d_iq, ctx.b.pre_tail[r.par], ctx.b.pre_tail[r.par ^ 1],
                   ctx.b.fm_in[r.buf], ctx.front);
    return hipGetLastError();
}

// the first decimator (m > 1 only): baseband -> fm_in[slot]
hipError_t launch_stage_predecim(const LaunchCtx& ctx, SlotRef r, const void* d_iq, bool u8, hipStream_t s) {
    const int m = ctx.d.m;
    if (u8) {
        const uchar2* p = static_cast<const uchar2*>(d_iq);
        return m == 4 ? launch_predecim<4, uchar2>(ctx, r, p, s) : launch_predecim<8, uchar2>(ctx, r, p, s);
    }
    const float2* p = static_cast<const float2*>(d_iq);
    return m == 4 ? launch_predecim<4, float2>(ctx, r, p, s) : launch_predecim<8, float2>(ctx, r, p, s);
}

// 1.024 / 2.048 MSa/s, tolerance mode: the first decimator inside the front end's kernel (k_front_pre_mfma) — the block then has no
// predecim stage and launch_stage_front takes the capture.  Not with a de-emphasised station (k_front_mfma's WU form stays on fm_in).
bool front_takes_capture(const LaunchCtx& ctx) {
    static const bool off = dev_env("FMD_NO_FRONT_PRE") != nullptr;      // (A/B hook)
    return ctx.d.m > 1 && ctx.fast && !ctx.any_deemph && !ctx.deemph_in_tile && ctx.d.n_fm_out % 1024 == 0 && !off && !ctx.split_front;
}

template <int M, typename InT>
static hipError_t launch_front_pre(const LaunchCtx& ctx, SlotRef r, const InT* d_iq, hipStream_t s, const SlotRef* pll) {
    using G = FrontPreGeom<M, sizeof(InT) == 2>;
    const Dims& d = ctx.d;
    PllFusedArgs pf{};
    if (pll) {
        const int nxt = (pll->buf + 1) % kSlots;
        pf = PllFusedArgs{(d.C + 31) / 32, ctx.b.pv_pl[pll->buf], ctx.b.pv_hist[pll->par], ctx.b.pv_hist[pll->par ^ 1], ctx.b.fo_pl[pll->buf], ctx.b.fo_pl[nxt],
                          ctx.b.pll_poly[pll->buf], ctx.b.pll_poly[nxt], ctx.b.state, ctx.loops, ctx.b.sparse_tab, ctx.b.spec_stats};
    }
    const unsigned grid = (unsigned)(d.n_fm_out / 1024 * d.C + pf.n_wg);
    if (pll) FMD_LAUNCH(r, true, true, (k_front_pre_mfma<M, InT, true>), dim3(grid), dim3(256), sizeof(float) * G::LDS_WORDS, s, d, d_iq, ctx.b.pre_tail[r.par], ctx.b.pre_tail[r.par ^ 1],
                        ctx.b.base_tail[r.par], ctx.b.base_tail[r.par ^ 1], ctx.b.fo_pl[r.buf], ctx.front.fm_gain, ctx.b.front_mfma + kFrontImgU4, ctx.b.front_mfma,
                        ctx.b.pv_pl[r.buf], ctx.b.sparse_tab, pf);
    else FMD_LAUNCH(r, true, true, (k_front_pre_mfma<M, InT, false>), dim3(grid), dim3(256), sizeof(float) * G::LDS_WORDS, s, d, d_iq, ctx.b.pre_tail[r.par], ctx.b.pre_tail[r.par ^ 1],
                    ctx.b.base_tail[r.par], ctx.b.base_tail[r.par ^ 1], ctx.b.fo_pl[r.buf], ctx.front.fm_gain, ctx.b.front_mfma + kFrontImgU4, ctx.b.front_mfma,
                    ctx.b.pv_pl[r.buf], ctx.b.sparse_tab, pf);
    return hipGetLastError();
}

// k_front on the 256 kSa/s stream: the capture itself (m == 1) or fm_in[slot]
hipError_t launch_stage_front(const LaunchCtx& ctx, SlotRef r, const void* d_iq, bool u8, hipStream_t s, const SlotRef* pll) {
    if (front_takes_capture(ctx)) {
        const int m = ctx.d.m;
        if (u8) {
            const uchar2* p = static_cast<const uchar2*>(d_iq);
            return m == 4 ? launch_front_pre<4, uchar2>(ctx, r, p, s, pll) : launch_front_pre<8, uchar2>(ctx, r, p, s, pll);
        }
        const float2* p = static_cast<const float2*>(d_iq);
        return m == 4 ? launch_front_pre<4, float2>(ctx, r, p, s, pll) : launch_front_pre<8, float2>(ctx, r, p, s, pll);
    }
    if (ctx.d.m > 1) {
        LaunchCtx c1 = ctx;
        c1.d.N = ctx.d.n_fm_in; c1.d.m = 1;
        if (ctx.fast) return launch_front<float>(c1, r, reinterpret_cast<const float*>(ctx.b.fm_in[r.buf]), s, pll);   // (phases: k_predecim<.., true>)
        return launch_front<float2>(c1, r, ctx.b.fm_in[r.buf], s);
    }
    if (u8) return launch_front<uchar2>(ctx, r, static_cast<const uchar2*>(d_iq), s, pll);
    return launch_front<float2>(ctx, r, static_cast<const float2*>(d_iq), s, pll);
}

static unsigned serial_waves(const Dims& d) { return (unsigned)((d.C + kWave - 1) / kWave); }

hipError_t launch_stage_deemph(const LaunchCtx& ctx, SlotRef r, hipStream_t s) {
    const Dims& d = ctx.d;
    const Buffers& b = ctx.b;
    if (ctx.fast) {   // a time constant beyond the in-tile form: in place on the plane (rows: kFoPad samples of history in front of the block)
        FMD_LAUNCH(r, true, false, k_deemphasis, dim3(serial_waves(d)), dim3(2 * kWave), 0, s, d, b.fo_pl[r.buf] + kFoPad, b.deemph, b.state, kFoPad + d.n_fm_out);
        const int ncol = d.n_fm_out / kSparseDec;
        FMD_LAUNCH(r, false, true, k_pilot_sums, dim3((unsigned)(d.C * ((ncol + 63) / 64))), dim3(256), 0, s, d, b.fo_pl[r.buf], b.pv_pl[r.buf], b.sparse_tab);
        return hipGetLastError();
    }
    FMD_LAUNCH(r, true, false, k_deemphasis, dim3(serial_waves(d)), dim3(2 * kWave), 0, s, d, b.fm_out[r.buf], b.deemph, b.state, d.n_fm_out);
    FMD_LAUNCH(r, false, true, k_hilbert, dim3((unsigned)(d.n_fm_out / 256 * d.C)), dim3(256), 0, s, d, b.fm_out[r.buf], b.fo_tail[r.par],
                       b.fo_tail[r.par ^ 1], b.fm_out_iq[r.buf], ctx.front);
    return hipGetLastError();
}

hipError_t launch_stage_power(const LaunchCtx& ctx, SlotRef r, hipStream_t s) {
    const Dims& d = ctx.d;
    if (effective_channels(d) <= 2816) {   // the batches whose step is this kernel's latency (and whose PLL launches hand over per wavefront)
        FMD_LAUNCH(r, true, true, k_pilot_power<true>, dim3(serial_waves(d)), dim3(kWave), 0, s, d, ctx.b.fm_out_iq[r.buf], ctx.b.pilot[r.buf], ctx.b.state,
                   ctx.loops, (int)S_PILOT_POWER0 + r.buf);
    } else {
        FMD_LAUNCH(r, true, true, k_pilot_power<false>, dim3(serial_waves(d)), dim3(kWave), 0, s, d, ctx.b.fm_out_iq[r.buf], ctx.b.pilot[r.buf], ctx.b.state,
                   ctx.loops, (int)S_PILOT_POWER0 + r.buf);
    }
    return hipGetLastError();
}

hipError_t launch_stage_pll(const LaunchCtx& ctx, SlotRef r, hipStream_t s) {
    const Dims& d = ctx.d;
    if (ctx.fast) {
        const int nxt = (r.buf + 1) % kSlots;
        const bool iq = ctx.b.fm_out_iq[r.buf] != nullptr;    // FMD_FLAG_KEEP_TAPS, or audio blocks that are not multiples of 256: the interleaved streams too
        // k_pll_sparse: 8 stations per wavefront, 4 wavefronts per workgroup.  In a block inside some station's start-up transient (r.warm)
        // k_pll_span runs behind it and takes those stations (fmd_kernels_fast.inc)
        FMD_LAUNCH(r, true, !iq && !r.warm, k_pll_sparse, dim3((unsigned)((d.C + 31) / 32)), dim3(4 * kWave), 0, s, d, ctx.b.pv_pl[r.buf], ctx.b.pv_hist[r.par], ctx.b.pv_hist[r.par ^ 1],
                   ctx.b.fo_pl[r.buf], ctx.b.fo_pl[nxt], ctx.b.pll_poly[r.buf], ctx.b.pll_poly[nxt], ctx.b.state, ctx.loops, ctx.b.sparse_tab, r.warm, ctx.b.spec_stats);
        if (r.warm)
            FMD_LAUNCH(r, false, !iq, k_pll_span, dim3((unsigned)((d.C + 3) / 4)), dim3(kWave), 0, s, d, ctx.b.fo_pl[r.buf], ctx.b.fo_pl[nxt],
                       ctx.b.pll_poly[r.buf], ctx.b.pll_poly[nxt], (float*)nullptr, ctx.b.state, ctx.loops, ctx.b.pilot_tab, ctx.b.span_tab, ctx.b.spec_stats, r.warm == 2 ? 1 : 0);
        if (iq) FMD_LAUNCH(r, false, false, k_poly_to_dt, dim3((unsigned)((size_t)d.n_fm_out * d.C / 256)), dim3(256), 0, s, d, ctx.b.pll_poly[r.buf], ctx.b.pll_dt[r.buf]);
        if (iq) FMD_LAUNCH(r, false, true, k_planes_to_iq, dim3((unsigned)(d.n_fm_out / 256 * d.C)), dim3(256), 0, s, d, ctx.b.fo_pl[r.buf], ctx.b.fm_out_iq[r.buf], ctx.front);
        return hipGetLastError();
    }
    // FMD_FLAG_KEEP_TAPS: the loop's traces, from the state the block starts from (ahead of the kernel that advances it)
    if (ctx.b.taps[r.buf]) hipLaunchKernelGGL(k_pll_taps, dim3((unsigned)((d.C + kWave - 1) / kWave)), dim3(kWave), 0, s, d, ctx.b.pilot[r.buf], ctx.b.state, ctx.loops,
                                              (int)S_PILOT_POWER0 + r.buf, tap_ptrs(ctx, r.buf));
    // (ctx.pll_unlocked_now: wavefronts ran out of lock in the last blocks the host has seen — the low-work kernel gives way to the time-parallel one,
    //  whose sequence form gets through such loops; the time-parallel kernel takes 16 lanes a station up to 4096 stations)
    if (d.C > ctx.pll_time_parallel_max_channels && !ctx.pll_unlocked_now) {
        FMD_LAUNCH(r, true, true, k_pilot_pll_pairs, dim3((unsigned)((d.C + kPllCh - 1) / kPllCh)), dim3(2 * kWave), 0, s, d, ctx.b.pilot[r.buf], ctx.b.pll_dt[r.buf],
                   ctx.b.state, ctx.loops, (int)S_PILOT_POWER0 + r.buf, ctx.b.spec_stats, ctx.b.pll_hint, ctx.pll_launch_no);
        return hipGetLastError();
    }
    unsigned int* chain = r.seq ? ctx.b.pll_chain : nullptr;
    if (effective_channels(d) <= ctx.pll_k16_max_channels || (ctx.pll_unlocked_now && effective_channels(d) <= 4096)) {
        FMD_LAUNCH(r, true, true, k_pilot_pll<16>, dim3((unsigned)((d.C + 3) / 4)), dim3(kWave), 0, s, d, ctx.b.pilot[r.buf], ctx.b.pll_dt[r.buf], ctx.b.state,
                   ctx.loops, (int)S_PILOT_POWER0 + r.buf, ctx.b.spec_stats, chain, r.seq, ctx.b.pll_hint, ctx.pll_launch_no);
    } else {
        FMD_LAUNCH(r, true, true, k_pilot_pll<8>, dim3((unsigned)((d.C + 7) / 8)), dim3(kWave), 0, s, d, ctx.b.pilot[r.buf], ctx.b.pll_dt[r.buf], ctx.b.state,
                   ctx.loops, (int)S_PILOT_POWER0 + r.buf, ctx.b.spec_stats, chain, r.seq, ctx.b.pll_hint, ctx.pll_launch_no);
    }
    return hipGetLastError();
}

// Tolerance mode, batches up to 6144 stations: same-box A/B +3 % at 4096 stations, but -3 % at 8192 and -4 % at 16384 (there the
// k_extract launches running back to back crowd k_front out: its launches take 1.7x as long); and in the exact mode the pilot
// PLL's launch chain sets the step, which k_extract launches without gaps between them slow down (-8 %).
static inline bool lmr_inline(const LaunchCtx& ctx) { return ctx.fast && ctx.d.n_est <= kLmrInlineMax && effective_channels(ctx.d) <= 6144; }
static inline int lmr_field(int par) { return par ? (int)S_LMR_PHASE_PREV : (int)S_LMR_PHASE_CUR; }   // state field holding P_k, k & 1 == par

template <int TA, bool FAST = false>
static void launch_extract_ta(const LaunchCtx& ctx, SlotRef r, hipStream_t s) {
    const Dims& d = ctx.d;
    const Buffers& b = ctx.b;
    if constexpr (!FAST) {
        if (ctx.fast) return launch_extract_ta<TA, true>(ctx, r, s);
    }
    if constexpr (FAST && TA == 256) {   // tolerance mode: the FIRs on the matrix cores
        // round 5: the mixers behind the FIRs, a wavefront per tile (fmd_kernels_bp.inc)
        const int tiles = d.n_audio / TA;
        // a workgroup = station x 4 nt tiles, nt per wavefront (the station's tap tables are staged once per workgroup): the largest nt that
        // still leaves 1536 workgroups for the chip (4 per CU: a round and a half)
        int nt = 1;
        // (nt travels in the low 16 bits of one kernel argument, the development switch in bit 16: block_size has no cap, so the search is
        //  clamped — ADVICE r5: at 8 bits a block of a million samples could wrap nt to 0)
        constexpr int kNtMax = 0x7fff;
        for (int v = (tiles + 3) / 4 < kNtMax ? (tiles + 3) / 4 : kNtMax; v >= 1; v--) if ((long)((tiles + 4 * v - 1) / (4 * v)) * d.C >= 1536) { nt = v; break; }
        if (const char* e = dev_env("FMD_BP_NT")) { const int v = atoi(e); if (v > 0 && v <= kNtMax) nt = v; }      // (development A/B)
        int grid_x = (tiles + 4 * nt - 1) / (4 * nt) * d.C;
        // round 6: two stations per workgroup (tables and the block edge's matrix fetched once for both) where a station is one workgroup anyway,
        // every station has the same cut-offs and the pairs still are 1536 workgroups; ctx.extract_pairing: 1 forces it (tests), 2 switches it off
        const bool one_wg_per_station = 4 * nt >= tiles;
        const bool pair = ctx.uniform_cutoffs && one_wg_per_station && 2 * tiles <= 4 * kNtMax && d.C >= 2 &&
                          (ctx.extract_pairing == 1 || (ctx.extract_pairing == 0 && (d.C + 1) / 2 >= 1536 && !dev_env("FMD_BP_NOPAIR")));
        if (pair) { nt = (2 * tiles + 3) / 4; grid_x = (d.C + 1) / 2; }
        if (dev_env("FMD_BP_NOEDGE")) nt |= 0x10000;                     // (development, timing only: the first tile's sums over the previous block skipped)
        if (pair) FMD_LAUNCH(r, true, true, k_extract_bp<2>, dim3((unsigned)grid_x), dim3(TA), 0, s, d, nt, b.fo_pl[r.buf], b.pll_poly[r.buf],
                   b.bp_tab, b.aud_idx, b.rds_bp_tab, b.bp_edge, b.mix,
                   b.state, b.audio[r.buf], b.rds[r.buf], b.lmr_est[r.par], b.lpr[r.buf], b.lmr[r.buf], ctx.keep_taps,
                   lmr_inline(ctx) ? b.lmr_est[r.par ^ 1] : (const float*)nullptr, lmr_field(r.par), lmr_field(r.par ^ 1), b.rds_pow[r.buf]);
        else FMD_LAUNCH(r, true, true, k_extract_bp<1>, dim3((unsigned)grid_x), dim3(TA), 0, s, d, nt, b.fo_pl[r.buf], b.pll_poly[r.buf],
                   b.bp_tab, b.aud_idx, b.rds_bp_tab, b.bp_edge, b.mix,
                   b.state, b.audio[r.buf], b.rds[r.buf], b.lmr_est[r.par], b.lpr[r.buf], b.lmr[r.buf], ctx.keep_taps,
                   lmr_inline(ctx) ? b.lmr_est[r.par ^ 1] : (const float*)nullptr, lmr_field(r.par), lmr_field(r.par ^ 1), b.rds_pow[r.buf]);
        return;
    }
    FMD_LAUNCH(r, true, true, (k_extract<TA, FAST>), dim3((unsigned)(d.n_audio / TA * d.C)), dim3(TA), 0, s, d, b.fm_out_iq[r.buf], b.pll_dt[r.buf],
                       b.iq_tail[r.par], b.dt_tail[r.par], b.iq_tail[r.par ^ 1], b.dt_tail[r.par ^ 1], b.b_lpr, b.b_lmr, ctx.rds_taps, b.mix,
                       b.state, b.audio[r.buf], b.rds[r.buf], b.lmr_est[r.par], b.lpr[r.buf], b.lmr[r.buf], ctx.keep_taps,
                       lmr_inline(ctx) ? b.lmr_est[r.par ^ 1] : (const float*)nullptr, lmr_field(r.par), lmr_field(r.par ^ 1));
}

hipError_t launch_stage_extract(const LaunchCtx& ctx, SlotRef r, hipStream_t s) {
    if (ctx.d.n_audio % 256 == 0) launch_extract_ta<256>(ctx, r, s);
    else launch_extract_ta<128>(ctx, r, s);
    if (!lmr_inline(ctx)) {   // P_{b+1} behind block b
        if (ctx.fast && ctx.d.n_est <= kLmrInlineMax)
            hipLaunchKernelGGL(k_lmr_phase_fast, dim3((unsigned)ctx.d.C), dim3(kWave), 0, s, ctx.d, ctx.b.lmr_est[r.par], ctx.b.state, lmr_field(r.par),
                               ctx.b.state + (size_t)lmr_field(r.par ^ 1) * ctx.d.C);
        else
            hipLaunchKernelGGL(k_lmr_phase, dim3(serial_waves(ctx.d)), dim3(kWave), 0, s, ctx.d, ctx.b.lmr_est[r.par], ctx.b.state, lmr_field(r.par),
                               ctx.b.state + (size_t)lmr_field(r.par ^ 1) * ctx.d.C);
    }
    return hipGetLastError();
}

// the value the reference's GetAudioLMRPhaseError() shows after the newest block (parity `par`): P_{b+1}, into out_row[C]
hipError_t launch_lmr_phase_peek(const LaunchCtx& ctx, int par, float* out_row, hipStream_t s) {
    if (ctx.fast && ctx.d.n_est <= kLmrInlineMax)
        hipLaunchKernelGGL(k_lmr_phase_fast, dim3((unsigned)ctx.d.C), dim3(kWave), 0, s, ctx.d, ctx.b.lmr_est[par], ctx.b.state, lmr_field(par), out_row);
    else
        hipLaunchKernelGGL(k_lmr_phase, dim3(serial_waves(ctx.d)), dim3(kWave), 0, s, ctx.d, ctx.b.lmr_est[par], ctx.b.state, lmr_field(par), out_row);
    return hipGetLastError();
}

#ifdef FMD_F_PROBE
}  // namespace fmd
extern "C" int fmd_debug_read_f_probe(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(fmd::g_f_probe), 16 * sizeof(unsigned long long));
}
namespace fmd {
#endif
#ifdef FMD_X_PROBE
}  // namespace fmd
extern "C" int fmd_debug_read_x_probe(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(fmd::g_x_probe), 16 * sizeof(unsigned long long));
}
extern "C" int fmd_debug_read_x_probe2(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(fmd::g_x_probe2), 16 * sizeof(unsigned long long));
}
namespace fmd {
#endif
}  // namespace fmd
// development: how many k_chain workgroups the runtime places on one CU (registers, LDS)
extern "C" int fmd_debug_chain_occupancy(void) {
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fmd::k_chain, 320, fmd::ChainLds::BYTES) != hipSuccess) return -1;
    return n;
}
namespace fmd {
#ifdef FMD_C_PROBE
}  // namespace fmd
extern "C" int fmd_debug_read_c_probe(unsigned long long* out32) {
    return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(fmd::g_c_probe), 32 * sizeof(unsigned long long));
}
extern "C" int fmd_debug_read_c_start(unsigned long long* out2048) {
    return (int)hipMemcpyFromSymbol(out2048, HIP_SYMBOL(fmd::g_c_start), 2048 * sizeof(unsigned long long));
}
namespace fmd {
#endif
#ifdef FMD_RDS_PROBE
}  // namespace fmd
extern "C" int fmd_debug_read_rds_probe(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(fmd::g_rds_probe), 16 * sizeof(unsigned long long));
}
namespace fmd {
#endif
// Tolerance mode, 256 kSa/s cf32 captures: the block's front end, pilot stage and extract stage as ONE launch (fmd_kernels_chain.inc); the
// RDS stage follows as before.  chain_possible: what the kernel's geometry needs of the configuration (the host adds what it needs of the
// block: no start-up block, no de-emphasis, one set of cut-offs for all stations — fmd_api.cpp).
bool chain_possible(const LaunchCtx& ctx) {
    const Dims& d = ctx.d;
    return ctx.fast && d.m == 1 && !ctx.keep_taps && !ctx.b.fm_out_iq[0] && d.n_fm_out % 1024 == 0 && d.n_audio % 256 == 0 && d.tail_base >= 64;
}
hipError_t launch_stage_chain(const LaunchCtx& ctx, SlotRef r, const void* d_iq, hipStream_t s) {
    const Dims& d = ctx.d;
    const Buffers& b = ctx.b;
    const int nxt = (r.buf + 1) % kSlots;
    ChainArgs a{};
    a.in = static_cast<const float2*>(d_iq); a.tail_in = b.base_tail[r.par]; a.tail_out = b.base_tail[r.par ^ 1];
    a.front_tab = b.front_mfma; a.sp = b.sparse_tab; a.fm_gain = ctx.front.fm_gain;
    a.fo_hist = b.fo_pl[r.buf]; a.fo_next = b.fo_pl[nxt];
    a.hist_in = b.pv_hist[r.par]; a.hist_out = b.pv_hist[r.par ^ 1];
    a.poly = b.pll_poly[r.buf]; a.poly_next = b.pll_poly[nxt];
    a.state = b.state; a.k = ctx.loops;
    a.bp_tab = b.bp_tab; a.aud_idx = b.aud_idx; a.rds_tab = b.rds_bp_tab; a.bp_edge = b.bp_edge; a.mixctl = b.mix;
    a.audio = b.audio[r.buf]; a.rds = b.rds[r.buf]; a.lmr_est = b.lmr_est[r.par];
    a.lmr_est_prev = lmr_inline(ctx) ? b.lmr_est[r.par ^ 1] : (const float*)nullptr;
    a.field_cur = lmr_field(r.par); a.field_prev = lmr_field(r.par ^ 1); a.rds_pow = b.rds_pow[r.buf];
    a.spec_stats = b.spec_stats;
    // (more than 64 KB of dynamic LDS has to be asked for; per call: a process may drive several devices)
    const hipError_t lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, ChainLds::BYTES);
    if (lds_ok != hipSuccess) return lds_ok;
    FMD_LAUNCH(r, true, true, k_chain, dim3((unsigned)((d.C + ChainGeom::G - 1) / ChainGeom::G)), dim3(320), ChainLds::BYTES, s, d, a);
    if (!lmr_inline(ctx)) {   // P_{b+1} behind block b (launch_stage_extract)
        if (ctx.fast && ctx.d.n_est <= kLmrInlineMax)
            hipLaunchKernelGGL(k_lmr_phase_fast, dim3((unsigned)ctx.d.C), dim3(kWave), 0, s, ctx.d, ctx.b.lmr_est[r.par], ctx.b.state, lmr_field(r.par),
                               ctx.b.state + (size_t)lmr_field(r.par ^ 1) * ctx.d.C);
        else
            hipLaunchKernelGGL(k_lmr_phase, dim3(serial_waves(ctx.d)), dim3(kWave), 0, s, ctx.d, ctx.b.lmr_est[r.par], ctx.b.state, lmr_field(r.par),
                               ctx.b.state + (size_t)lmr_field(r.par ^ 1) * ctx.d.C);
    }
    return hipGetLastError();
}

hipError_t launch_stage_rds(const LaunchCtx& ctx, SlotRef r, hipStream_t s) {
    const Dims& d = ctx.d;
    const Buffers& b = ctx.b;
    if (ctx.fast) {
        const bool partials = d.n_audio % 256 == 0;       // k_extract_bp ran and left the block's power as 2 partial sums per tile
        static const bool two_waves = dev_env("FMD_RDS_TWO_WAVES") != nullptr;      // (A/B hook: the two-wavefront form)
        if (partials && !two_waves) {      // the loop split over mixer, clock and dump wavefronts (fmd_kernels_fast.inc)
            FMD_LAUNCH(r, true, true, k_rds_sync3, dim3(serial_waves(d)), dim3(5 * kWave), 0, s, d, b.rds[r.buf], b.state, ctx.loops, b.rds_sym[r.buf],
                       b.rds_raw_sym[r.buf], b.rds_count[r.buf], b.rds_bytes[r.buf], b.rds_bytes_count[r.buf], ctx.bytes_cap, ctx.keep_taps,
                       b.rds_pow[r.buf], 2 * (d.n_audio / 256));
            return hipGetLastError();
        }
        FMD_LAUNCH(r, true, true, k_rds_sync<true>, dim3(serial_waves(d)), dim3(2 * kWave), 0, s, d, b.rds[r.buf], b.state, ctx.loops, b.rds_sym[r.buf],
                   b.rds_raw_sym[r.buf], b.rds_count[r.buf], b.rds_bytes[r.buf], b.rds_bytes_count[r.buf], ctx.bytes_cap, ctx.keep_taps,
                   partials ? b.rds_pow[r.buf] : (const float*)nullptr, 2 * (d.n_audio / 256), TapPtrs{});
    } else {
        FMD_LAUNCH(r, true, true, k_rds_sync<false>, dim3(serial_waves(d)), dim3(2 * kWave), 0, s, d, b.rds[r.buf], b.state, ctx.loops, b.rds_sym[r.buf],
                   b.rds_raw_sym[r.buf], b.rds_count[r.buf], b.rds_bytes[r.buf], b.rds_bytes_count[r.buf], ctx.bytes_cap, ctx.keep_taps, (const float*)nullptr, 0,
                   tap_ptrs(ctx, r.buf));
    }
    return hipGetLastError();
}

template <typename InT, int TT = 512>
static hipError_t prepare_front() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_front<InT, TT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(float) * FrontGeom<TT>::LDS_FLOATS));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_mfma<InT, TT, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * FrontGeomM<TT, 0>::LDS_FLOATS) + 49152);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_mfma<InT, TT, kDeemphWarmup, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * FrontGeomM<TT, kDeemphWarmup>::LDS_FLOATS) + 49152);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_mfma<InT, TT, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * FrontGeomM<TT, 0>::LDS_FLOATS) + 49152);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_mfma<InT, TT, kDeemphWarmup, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * FrontGeomM<TT, kDeemphWarmup>::LDS_FLOATS) + 49152);
    if (e != hipSuccess) return e;
    return e;
}

hipError_t prepare_kernels() {
    hipError_t e;
    if ((e = prepare_front<float2>()) != hipSuccess) return e;
    if ((e = prepare_front<float2, 1024>()) != hipSuccess) return e;
    if ((e = prepare_front<uchar2>()) != hipSuccess) return e;
    if ((e = prepare_front<uchar2, 1024>()) != hipSuccess) return e;
    if ((e = prepare_front<float>()) != hipSuccess) return e;
    if ((e = prepare_front<float, 1024>()) != hipSuccess) return e;
    return hipSuccess;
}

// k_front always runs at 256 kSa/s (the first decimator keeps its own 64 samples); the tolerance mode keeps the history its in-tile
// de-emphasis needs, whether or not a channel uses it now
int front_tail_len(int m, bool fast) { (void)m; return fast ? FrontGeomM<512, kDeemphWarmup>::TAIL : FrontGeom<>::TAIL; }

hipError_t launch_reset_state(const LaunchCtx& ctx, hipStream_t stream) {
    hipLaunchKernelGGL(k_reset, dim3((unsigned)((ctx.d.C + 255) / 256)), dim3(256), 0, stream, ctx.d, ctx.b.state);
    return hipGetLastError();
}

}  // namespace fmd
