// Device math for the FM demodulation kernels.
//
// Everything here is written so that every operation rounds exactly once, in a fixed order:
// the translation units that include this header are compiled with -ffp-contract=off and all
// fused multiply-adds are explicit fmaf() calls.  The orders are those of the reference's
// AVX2+FMA build (see DESIGN.md "Arithmetic contract"), so the GPU results can be compared
// bit-for-bit with the CPU path.
//
// The header also compiles as plain host C++ (FMD_HD empty) so tests can check fmd_atan2f
// against the host libm over large random sweeps without a GPU.
#pragma once

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define FMD_HD __host__ __device__ __forceinline__
#else
#define FMD_HD static inline
#endif

namespace fmd {

FMD_HD float bits_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
FMD_HD uint32_t f32_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}

constexpr uint32_t kPiBits = 0x40490fdbu;         // (float)M_PI
constexpr uint32_t kTwoPiBits = 0x40c90fdbu;      // 2*pi
constexpr uint32_t kHalfPiBits = 0x3fc90fdbu;     // pi/2
constexpr uint32_t kTwoOverPiBits = 0x3f22f983u;  // 2/pi
constexpr uint32_t kPredHalfBits = 0x3effffffu;   // nextafter(0.5, 0)

// ---------------------------------------------------------------------------------------------
// atan2f / atanf — the FDLIBM float algorithm (Sun Microsystems; e_atan2f.c / s_atanf.c) that the
// host C library the reference links against implements (glibc 2.35, sysdeps/ieee754/flt-32).
// The reference calls std::atan2 in its discriminator (reference src/fm_demod/fm_demod.cpp:40),
// pilot PLL (broadcast_fm_demod.cpp:450), L-R phase tracker (:502) and BPSK synchroniser
// (bpsk_synchroniser.cpp:159); reproducing the same sequence of IEEE operations keeps those
// feedback loops bit-identical on the GPU.  The argument reduction is written branch-free
// (one select per range, ONE division) so a wavefront never diverges on it.
// ---------------------------------------------------------------------------------------------
FMD_HD float atanf_core(float x) {
    const uint32_t hx = f32_bits(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    if (ix >= 0x4c000000u) {  // |x| >= 2^25 (or NaN)
        if (ix > 0x7f800000u) return x + x;
        const float r = bits_f32(0x3fc90fdau) + bits_f32(0x33a22168u);
        return neg ? -r : r;
    }
    const float ax = fabsf(x);
    // id: -1 (|x| < 7/16), 0 (< 11/16), 1 (< 19/16), 2 (< 39/16), 3 (else)
    const bool r_small = ix < 0x3ee00000u;
    const bool r0 = ix < 0x3f300000u;
    const bool r1 = ix < 0x3f980000u;
    const bool r2 = ix < 0x401c0000u;
    // numerators / denominators of the reduced argument, each rounded as the C expression would
    const float two_x = 2.0f * ax;
    const float n0 = two_x - 1.0f, d0 = 2.0f + ax;
    const float n1 = ax - 1.0f, d1 = ax + 1.0f;
    const float n2 = ax - 1.5f, d2 = 1.0f + 1.5f * ax;
    float num = r2 ? n2 : -1.0f, den = r2 ? d2 : ax;
    num = r1 ? n1 : num; den = r1 ? d1 : den;
    num = r0 ? n0 : num; den = r0 ? d0 : den;
    float hi = r2 ? bits_f32(0x3f7b985eu) : bits_f32(0x3fc90fdau);
    float lo = r2 ? bits_f32(0x33140fb4u) : bits_f32(0x33a22168u);
    hi = r1 ? bits_f32(0x3f490fdau) : hi; lo = r1 ? bits_f32(0x33222168u) : lo;
    hi = r0 ? bits_f32(0x3eed6338u) : hi; lo = r0 ? bits_f32(0x31ac3769u) : lo;
    const float xr = r_small ? x : (num / den);
    if (r_small && ix < 0x31000000u) return x;  // |x| < 2^-29
    const float z = xr * xr;
    const float w = z * z;
    // odd / even halves of sum aT[i] z^(i+1), plain multiply-add chain (no fusing)
    float s1 = bits_f32(0x3d4bda59u) + w * bits_f32(0x3c8569d7u);   // aT[8] + w*aT[10]
    s1 = bits_f32(0x3d886b35u) + w * s1;                             // aT[6]
    s1 = bits_f32(0x3dba2e6eu) + w * s1;                             // aT[4]
    s1 = bits_f32(0x3e124925u) + w * s1;                             // aT[2]
    s1 = bits_f32(0x3eaaaaabu) + w * s1;                             // aT[0]
    s1 = z * s1;
    float s2 = bits_f32(0xbd6ef16bu) + w * bits_f32(0xbd15a221u);   // aT[7] + w*aT[9]
    s2 = bits_f32(0xbd9d8795u) + w * s2;                             // aT[5]
    s2 = bits_f32(0xbde38e38u) + w * s2;                             // aT[3]
    s2 = bits_f32(0xbe4ccccdu) + w * s2;                             // aT[1]
    s2 = w * s2;
    const float p = xr * (s1 + s2);
    if (r_small) return xr - p;
    const float r = hi - ((p - lo) - xr);
    return neg ? -r : r;
}

// The algorithm exactly as published, special cases included (NaN, zeros, infinities, |y/x| beyond 2^+-60).
FMD_HD float fmd_atan2f_full(float y, float x) {
    const uint32_t hx = f32_bits(x), hy = f32_bits(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const float pi = bits_f32(kPiBits), pi_o_2 = bits_f32(kHalfPiBits), pi_lo = bits_f32(0xb3bbbd2eu);
    const float tiny = 1.0e-30f;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;  // NaN
    if (hx == 0x3f800000u) return atanf_core(y);              // x == 1
    const uint32_t m = ((hy >> 31) & 1u) | ((hx >> 30) & 2u);
    if (iy == 0u) {  // y == 0
        if (m < 2u) return y;
        return (m == 2u) ? (pi + tiny) : (-pi - tiny);
    }
    if (ix == 0u) return (hy >> 31) ? (-pi_o_2 - tiny) : (pi_o_2 + tiny);  // x == 0
    if (ix == 0x7f800000u) {  // x == inf
        const float pi_o_4 = bits_f32(0x3f490fdbu);
        if (iy == 0x7f800000u) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000u) return (hy >> 31) ? (-pi_o_2 - tiny) : (pi_o_2 + tiny);
    const int32_t k = ((int32_t)iy - (int32_t)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if ((hx >> 31) && k < -60) z = 0.0f;
    else z = atanf_core(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return bits_f32(f32_bits(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// atanf of a non-negative finite-or-infinite argument, same operations as atanf_core but with every range
// decision turned into a select: ONE division, no branches, so a wavefront never diverges on it.
FMD_HD float atanf_pos_branchless(float ax) {
    const uint32_t ix = f32_bits(ax);
    const bool r_small = ix < 0x3ee00000u;
    const bool r0 = ix < 0x3f300000u;
    const bool r1 = ix < 0x3f980000u;
    const bool r2 = ix < 0x401c0000u;
    const float two_x = 2.0f * ax;
    const float n0 = two_x - 1.0f, d0 = 2.0f + ax;
    const float n1 = ax - 1.0f, d1 = ax + 1.0f;
    const float n2 = ax - 1.5f, d2 = 1.0f + 1.5f * ax;
    // two-level select tree (r0 implies r1 implies r2): the selects sit on the latency-critical path
    const float num_lo = r0 ? n0 : n1, den_lo = r0 ? d0 : d1;
    const float num_hi = r2 ? n2 : -1.0f, den_hi = r2 ? d2 : ax;
    const float num = r1 ? num_lo : num_hi, den = r1 ? den_lo : den_hi;
    const float hi_lo = r0 ? bits_f32(0x3eed6338u) : bits_f32(0x3f490fdau), lo_lo = r0 ? bits_f32(0x31ac3769u) : bits_f32(0x33222168u);
    const float hi_hi = r2 ? bits_f32(0x3f7b985eu) : bits_f32(0x3fc90fdau), lo_hi = r2 ? bits_f32(0x33140fb4u) : bits_f32(0x33a22168u);
    const float hi = r1 ? hi_lo : hi_hi, lo = r1 ? lo_lo : lo_hi;
    const float quot = num / den;
    const float xr = r_small ? ax : quot;
    const float z = xr * xr;
    const float w = z * z;
    float s1 = bits_f32(0x3d4bda59u) + w * bits_f32(0x3c8569d7u);
    s1 = bits_f32(0x3d886b35u) + w * s1;
    s1 = bits_f32(0x3dba2e6eu) + w * s1;
    s1 = bits_f32(0x3e124925u) + w * s1;
    s1 = bits_f32(0x3eaaaaabu) + w * s1;
    s1 = z * s1;
    float s2 = bits_f32(0xbd6ef16bu) + w * bits_f32(0xbd15a221u);
    s2 = bits_f32(0xbd9d8795u) + w * s2;
    s2 = bits_f32(0xbde38e38u) + w * s2;
    s2 = bits_f32(0xbe4ccccdu) + w * s2;
    s2 = w * s2;
    const float p = xr * (s1 + s2);
    const float r = r_small ? (xr - p) : (hi - ((p - lo) - xr));
    // |x| < 2^-29: atan x = x;  |x| >= 2^25: atanhi[3]+atanlo[3] = fl(pi/2).  Both conditions and both values are known
    // long before r, so they cost one select on the critical path instead of two.
    const bool edge = (ix < 0x31000000u) | (ix >= 0x4c000000u);
    const float edge_val = (ix < 0x31000000u) ? ax : bits_f32(kHalfPiBits);
    return edge ? edge_val : r;
}

// atan2f used by the kernels.  For finite non-zero x and y (everything a live signal produces) the published
// algorithm reduces to: z = atanf(|y/x|), then one of four sign/quadrant fix-ups — evaluated here with
// selects.  Its |y/x| > 2^60 and |y|/x < -2^-60 shortcuts give the same floats as the general path (the
// shortcut constants equal atanf's own saturation values after rounding; checked exhaustively around the
// thresholds by tests/test_capi_cpu.py), and x == 1 is atanf(y) which is odd-symmetric in this algorithm.
// Zeros, infinities and NaNs take the (rare, wave-uniformly skipped) full version.
FMD_HD float fmd_atan2f(float y, float x) {
    const uint32_t hx = f32_bits(x), hy = f32_bits(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const bool special = (ix == 0u) | (iy == 0u) | (ix >= 0x7f800000u) | (iy >= 0x7f800000u);
    const float pi = bits_f32(kPiBits), pi_lo = bits_f32(0xb3bbbd2eu);
    const float z = atanf_pos_branchless(fabsf(y / x));
    // quadrant fix-up (a bit-field-insert form of the sign transplant measured 1.7 % slower than these selects)
    const float zl = z - pi_lo;
    const bool sx = (hx >> 31) != 0;
    const bool sy = (hy >> 31) != 0;
    const float rpos = sy ? bits_f32(f32_bits(z) ^ 0x80000000u) : z;
    const float rneg = sy ? (zl - pi) : (pi - zl);
    float r = sx ? rneg : rpos;
    if (special) r = fmd_atan2f_full(y, x);
    return r;
}

// ---------------------------------------------------------------------------------------------
// Short forms for the latency-bound serial loops (pilot PLL, BPSK synchroniser), see fmd_kernels.hip pll_step_locked.
// Each computes EXACTLY the value of its general form whenever its stated precondition holds; the loops run a span of
// samples with the short forms, record the preconditions off the critical path, and replay the span with the general
// forms (wave-uniformly) if any lane violated one.  Nothing is approximated.
// ---------------------------------------------------------------------------------------------

// y / x by the IEEE expansion (rcp, two refinements of the reciprocal and quotient, final residual correction) without
// its operand scaling and special-value fix-up steps: the correctly rounded quotient whenever neither would act, i.e.
// for normal operands whose quotient and residuals stay normal (the caller checks ranges).
FMD_HD float div_unscaled(float y, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(x);
#else
    float r = 1.0f / x;
#endif
    const float e0 = fmaf(-x, r, 1.0f);
    r = fmaf(e0, r, r);
    float q = y * r;
    const float e1 = fmaf(-x, q, y);
    q = fmaf(e1, r, q);
    const float e2 = fmaf(-x, q, y);
    return fmaf(e2, r, q);
}

// ---------------------------------------------------------------------------------------------
// Throughput form of atan2f for the discriminator (k_front: two of these per output sample, every range equally likely).
// Same operations as fmd_atan2f, arranged for the fewest issued instructions rather than the shortest dependent chain:
//  * the five argument ranges differ only in constants, so they come from a 5-row table indexed through a byte map of
//    bits(|y/x|) >> 18 (all four range thresholds are multiples of 2^18):  num = fma(cn, a, bn), den = ce*a + bd,
//    result = hi - ((p - lo) - num/den).  Row 0 (|a| < 7/16: num = a, den = 1, hi = lo = 0) reproduces 'a - p', and the
//    |a| < 2^-29 early return as well, because there p is far below half an ulp of a.
//  * num/den never needs the IEEE division's operand scaling or special-value fix-up (den in [1, 2^25), num 0 or in
//    [2^-24, 2^25), or num = a, den = 1), so it is the bare expansion div_unscaled();
//  * |a| >= 2^25 (including the infinity of a zero x) and the sign/quadrant fix-up are two selects and a sign transplant.
// The table lives in LDS on the device (AtanTable filled by atan_table_fill) and in ordinary memory on the host.
// ---------------------------------------------------------------------------------------------
struct alignas(32) AtanTable {
    float cls[5][8];   // cn, bn, ce, bd, hi, lo, -, -
    uint8_t idx[96];   // byte offset of the row for clamp(bits >> 18 & 0x1fff, kAtanIdxLo, kAtanIdxHi) - kAtanIdxLo
};
constexpr uint32_t kAtanIdxLo = 0x0fb7u;     // anything below 0x3ee00000 (7/16)
constexpr uint32_t kAtanIdxHi = 0x1007u;     // 0x401c0000 (39/16) and above
constexpr uint32_t kAtanIdxHuge = 0x1300u;   // 0x4c000000 (2^25) and above, infinity included
constexpr int kAtanTableWords = (int)(sizeof(AtanTable) / 4);

// word w of the table image (the idx bytes packed little-endian four to a word)
FMD_HD uint32_t atan_table_word(int w) {
    if (w < 40) {
        const int row = w >> 3, col = w & 7;
        const uint32_t one = 0x3f800000u, two = 0x40000000u, m_one = 0xbf800000u;
        switch (row * 8 + col) {
            case 0: return one;  case 3: return one;                                  // |a| < 7/16: a / 1
            case 8: return two;  case 9: return m_one; case 10: return one; case 11: return two;      // (2a - 1) / (2 + a)
            case 12: return 0x3eed6338u; case 13: return 0x31ac3769u;
            case 16: return one; case 17: return m_one; case 18: return one; case 19: return one;     // (a - 1) / (a + 1)
            case 20: return 0x3f490fdau; case 21: return 0x33222168u;
            case 24: return one; case 25: return 0xbfc00000u; case 26: return 0x3fc00000u; case 27: return one;  // (a - 1.5) / (1 + 1.5a)
            case 28: return 0x3f7b985eu; case 29: return 0x33140fb4u;
            case 33: return m_one; case 34: return one;                               // -1 / a
            case 36: return 0x3fc90fdau; case 37: return 0x33a22168u;
            default: return 0u;
        }
    }
    uint32_t v = 0;
    for (int b = 0; b < 4; b++) {
        const uint32_t e = kAtanIdxLo + (uint32_t)((w - 40) * 4 + b);
        const uint32_t row = (e >= 0x1007u) ? 4u : (e >= 0x0fe6u) ? 3u : (e >= 0x0fccu) ? 2u : (e >= 0x0fb8u) ? 1u : 0u;
        v |= (row * 32u) << (8 * b);
    }
    return v;
}
FMD_HD void atan_table_fill(AtanTable* t, int first, int stride) {
    uint32_t* w = reinterpret_cast<uint32_t*>(t);
    for (int i = first; i < kAtanTableWords; i += stride) w[i] = atan_table_word(i);
}

struct alignas(16) AtanRowA { float cn, bn, ce, bd; };
struct alignas(8) AtanRowB { float hi, lo; };

// kSmallInts: x and y are small integers held in floats (u8 IQ minus 127): no NaN, infinity or -0 can occur, and an
// operand is exactly +0 once in 256 samples.  With a single zero the table form below is already right (y/x = +-inf ->
// pi/2 with y's sign; +-0/x -> +-0, or +-pi through the x < 0 reflection), so only 0/0 needs the published special case —
// otherwise 40 % of the wavefronts of a u8 stream at 256 kSa/s would detour through it.
template <bool kSmallInts = false>
FMD_HD float fmd_atan2f_table(float y, float x, const AtanTable* t) {
    const uint32_t hx = f32_bits(x), hy = f32_bits(y);
#if defined(__HIP_DEVICE_COMPILE__)
    // zero, infinity or NaN in either operand
    const bool special = kSmallInts ? ((hx | hy) == 0u) : (__builtin_amdgcn_classf(x, 0x267) | __builtin_amdgcn_classf(y, 0x267));
#else
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    const bool special = kSmallInts ? ((hx | hy) == 0u) : ((ix == 0u) | (iy == 0u) | (ix >= 0x7f800000u) | (iy >= 0x7f800000u));
#endif
    const float q = y / x;
    const float a = fabsf(q);
    uint32_t e = (f32_bits(q) >> 18) & 0x1fffu;
    const bool huge = e >= kAtanIdxHuge;
    e = e < kAtanIdxLo ? kAtanIdxLo : e;
    e = e > kAtanIdxHi ? kAtanIdxHi : e;
    const char* row = reinterpret_cast<const char*>(t->cls) + t->idx[e - kAtanIdxLo];
    const AtanRowA ra = *reinterpret_cast<const AtanRowA*>(row);
    const AtanRowB rb = *reinterpret_cast<const AtanRowB*>(row + 16);
    const float num = fmaf(ra.cn, a, ra.bn);
    const float den = ra.ce * a + ra.bd;
    const float xr = div_unscaled(num, den);
    const float z = xr * xr;
    const float w = z * z;
    float s1 = bits_f32(0x3d4bda59u) + w * bits_f32(0x3c8569d7u);
    s1 = bits_f32(0x3d886b35u) + w * s1;
    s1 = bits_f32(0x3dba2e6eu) + w * s1;
    s1 = bits_f32(0x3e124925u) + w * s1;
    s1 = bits_f32(0x3eaaaaabu) + w * s1;
    s1 = z * s1;
    float s2 = bits_f32(0xbd6ef16bu) + w * bits_f32(0xbd15a221u);
    s2 = bits_f32(0xbd9d8795u) + w * s2;
    s2 = bits_f32(0xbde38e38u) + w * s2;
    s2 = bits_f32(0xbe4ccccdu) + w * s2;
    s2 = w * s2;
    const float p = xr * (s1 + s2);
    float at = rb.hi - ((p - rb.lo) - xr);
    at = huge ? bits_f32(kHalfPiBits) : at;
    // x < 0: pi - (at - pi_lo); then the sign of y (the published m = 1 and m = 3 cases are the negations of m = 0 and m = 2)
    const float refl = bits_f32(kPiBits) - (at - bits_f32(0xb3bbbd2eu));
    const float mag = (hx >> 31) ? refl : at;
    float r = bits_f32((f32_bits(mag) & 0x7fffffffu) | (hy & 0x80000000u));
    if (special) r = fmd_atan2f_full(y, x);
    return r;
}

// std::round as the reference build inlines it: trunc(x + copysign(pred(0.5), x))
FMD_HD float round_half_away(float x) { return truncf(x + copysignf(bits_f32(kPredHalfBits), x)); }

// reference src/dsp/clamp.h:3-8
FMD_HD float clampf(float x, float lo, float hi) {
    // == (x > lo ? x : lo), then (y < hi ? y : hi) for every non-NaN x (vmaxss / vminss in the reference build)
    return fminf(fmaxf(x, lo), hi);
}

// chebyshev sine, reference src/dsp/simd/chebyshev_sine.h:13-41 / :78-104.
FMD_HD float cheb_poly(float z) {
    float p = fmaf(3.20396066f, z, -14.07150173f);
    p = fmaf(p, z, 38.50016403f);
    p = fmaf(p, z, -67.07687378f);
    p = fmaf(p, z, 64.83583069f);
    p = fmaf(p, z, -25.13274193f);
    return p;
}
// scalar call sites (PLL loops): ((z - 1/4) x) g(z)
FMD_HD float cheb_sine_scalar(float x) {
    const float z = x * x;
    return ((z - 0.25f) * x) * cheb_poly(z);
}
// vector call site (harmonic mixer): (x g(z)) (z + -1/4)
FMD_HD float cheb_sine_vector(float x) {
    const float z = x * x;
    return (x * cheb_poly(z)) * (z + -0.25f);
}

// ---------------------------------------------------------------------------------------------
// FMD_FLAG_FAST_MATH — the tolerance mode (BASELINE north star: audio / L-R / RDS symbols within 1e-4 RMS of the
// reference, RDS bits identical).  Same signal flow as the reference, cheaper arithmetic: a minimax arctangent instead of
// libm's, the hardware sine/cosine (input in turns — the unit the reference's NCO phases already have) instead of the
// chebyshev polynomial, free summation order.  Measured accuracies: tests/test_gpu_fast.py::test_fast_math_primitives.
// ---------------------------------------------------------------------------------------------

// atan(a) for a in [0, 1]: a * P(a^2), degree-15 odd minimax (8 terms), max error 1.3e-7 rad in float arithmetic
FMD_HD float fast_atan_unit(float a) {
    const float z = a * a;
    float p = fmaf(-0.00405453285202384f, z, 0.021862823516130447f);
    p = fmaf(p, z, -0.055912118405103683f);
    p = fmaf(p, z, 0.09642180055379868f);
    p = fmaf(p, z, -0.13908621668815613f);
    p = fmaf(p, z, 0.19946563243865967f);
    p = fmaf(p, z, -0.33329859375953674f);
    p = fmaf(p, z, 0.9999993443489075f);
    return p * a;
}

// atan2(y, x), every quadrant, max error ~3e-7 rad; atan2(+0, +0) = 0 like libm's (the u8 path produces exact zeros).
// ~21 instructions, no division (one reciprocal), no table, no divergent special cases: NaN in -> NaN out.
FMD_HD float fast_atan2f(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1.0e-37f), mn = fminf(ax, ay);
#if defined(__HIP_DEVICE_COMPILE__)
    const float a = mn * __builtin_amdgcn_rcpf(mx);
#else
    const float a = mn / mx;
#endif
    float r = fast_atan_unit(a);
    r = (ay > ax) ? (bits_f32(kHalfPiBits) - r) : r;
    r = (f32_bits(x) >> 31) ? (bits_f32(kPiBits) - r) : r;
    return bits_f32((f32_bits(r) & 0x7fffffffu) | (f32_bits(y) & 0x80000000u));
}

// In TURNS (atan2 / 2 pi, in (-1/2, 1/2]): the polynomial's coefficients carry the 1 / 2 pi, the octant folding uses 1/4 and 1/2.
// For phases that are only ever differenced and wrapped (the discriminator: wrap = x - rint(x)) or added to an NCO phase in turns.
// Six coefficients (minimax on [0, 1], tools/proto/atan_fit.py): 2.8e-7 turns (1.7e-6 rad) — two discriminator phases differenced
// and scaled by the FM gain put at most 2e-6 into fm_out, against the 1e-4 RMS the mode promises and the 1e-5 it measures.
FMD_HD float fast_atan2_turns(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
#if defined(__HIP_DEVICE_COMPILE__)
    // (fminf / fmaxf on |x| compile to a canonicalising v_max per operand in front of the v_min / v_max3: two of this function's 21
    // instructions; the operands are finite samples, the bare instructions do)
    float mx, mn;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(mx) : "v"(x), "v"(y), "v"(1.0e-37f));
    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
    const float a = mn * __builtin_amdgcn_rcpf(mx);
#else
    const float mx = fmaxf(fmaxf(ax, ay), 1.0e-37f), mn = fminf(ax, ay);
    const float a = mn / mx;
#endif
    const float z = a * a;
    float p = fmaf(-0.011719091795384884f * 0.15915494309189535f, z, 0.05264724791049957f * 0.15915494309189535f);
    p = fmaf(p, z, -0.11642639338970184f * 0.15915494309189535f);
    p = fmaf(p, z, 0.19354034960269928f * 0.15915494309189535f);
    p = fmaf(p, z, -0.33262282609939575f * 0.15915494309189535f);
    p = fmaf(p, z, 0.9999772310256958f * 0.15915494309189535f);
    float r = p * a;
    r = (ay > ax) ? (0.25f - r) : r;
    r = (f32_bits(x) >> 31) ? (0.5f - r) : r;
    return bits_f32((f32_bits(r) & 0x7fffffffu) | (f32_bits(y) & 0x80000000u));
}

// sin(2 pi t), cos(2 pi t) for t in turns (|t| <= 256: no range reduction needed for the phases of this chain)
FMD_HD float fast_sin_turns(float t) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sinf(t);
#else
    return (float)sin(6.283185307179586 * (double)t);
#endif
}
FMD_HD float fast_cos_turns(float t) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_cosf(t);
#else
    return (float)cos(6.283185307179586 * (double)t);
#endif
}

}  // namespace fmd
