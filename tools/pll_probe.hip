// Development tool: times variants of the pilot-PLL serial loop on a locked synthetic pilot (not part of the product).
//   pll_probe [C=4096] [blocks=12]
#include "../fm-radio_amd/csrc/fmd_kernels.hip"
#include <cstdio>
#include <vector>
using namespace fmd;

struct Checks { float clamp_max; float half_min; uint32_t x_max, t_max; };

// variant step: verification accumulated in VALU registers only (no v_cmp -> SALU hazards inside the loop)
__device__ __forceinline__ float pll_step_v1(PllState& s, float p, float q, const LoopCoeffs& k, Checks& ck) {
    const float Ts = 1.0f / 128000.0f;
    const float KTsI = 0.1f * Ts;
    const float t0 = fmaf(s.lx1, k.pll_b0, s.ly1 * k.pll_a0);
    const float t1 = fmaf(s.err, k.pll_b1, 0.0f);
    const float lpf = (0.0f + t0) + t1;
    s.lx1 = s.err; s.ly1 = lpf;
    const float P = lpf * 0.01f;
    s.integ = fmaf(s.err, KTsI, s.integ);
    const float PI_error = s.integ + P;
    ck.clamp_max = fmaxf(fmaxf(ck.clamp_max, fabsf(s.integ)), fabsf(PI_error));
    const float freq = fmaf(PI_error, -100.0f, -19000.0f);
    const float yy = fmaf(freq, Ts, s.tph);
    ck.half_min = fminf(ck.half_min, fabsf(fabsf(yy) - 0.5f));
    s.tph = yy - rintf(yy);
    const float dc = s.tph + 0.25f;
    ck.half_min = fminf(ck.half_min, fabsf(fabsf(dc) - 0.5f));
    const float dt_cos = dc - rintf(dc);
    const float ps = cheb_sine_scalar(s.tph);
    const float pc = cheb_sine_scalar(dt_cos);
    const float res_im = fmaf(ps, p, q * pc);
    const float res_re = fmaf(p, pc, -(q * ps));
    const float t = div_unscaled(res_im, res_re);
    const uint32_t hx = f32_bits(res_re), it = f32_bits(t) & 0x7fffffffu;
    ck.x_max = max(ck.x_max, hx - 0x20800000u);
    ck.t_max = max(ck.t_max, it - 0x31000000u);
    const float z = t * t;
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
    s.err = t - t * (s1 + s2);
    return s.tph;
}

// all loop constants in VGPRs (opaque to the compiler): a 32-bit literal in the instruction stream costs a lone wave
// ~2.7 cycles of issue per instruction
struct PllConstsP {
    float b0, a0, b1, c001, ktsi, m100, m19000, ts, q25, mq25, c5, c4, c3, c2, c1, c0;
    float a10, a8, a6, a4, a2, a0t, a9, a7, a5, a3, a1;
    uint32_t absmask, xlo, tlo;
};
#define OPQ(dst, val) { float t_ = (val); asm volatile("" : "+v"(t_)); dst = t_; }
#define OPQU(dst, val) { uint32_t t_ = (val); asm volatile("" : "+v"(t_)); dst = t_; }
__device__ __forceinline__ PllConstsP make_consts(const LoopCoeffs& k) {
    PllConstsP c;
    OPQ(c.b0, k.pll_b0) OPQ(c.a0, k.pll_a0) OPQ(c.b1, k.pll_b1) OPQ(c.c001, 0.01f) OPQ(c.ktsi, 0.1f * (1.0f / 128000.0f)) OPQ(c.m100, -100.0f)
    OPQ(c.m19000, -19000.0f) OPQ(c.ts, 1.0f / 128000.0f) OPQ(c.q25, 0.25f) OPQ(c.mq25, -0.25f)
    OPQ(c.c5, 3.20396066f) OPQ(c.c4, -14.07150173f) OPQ(c.c3, 38.50016403f) OPQ(c.c2, -67.07687378f) OPQ(c.c1, 64.83583069f) OPQ(c.c0, -25.13274193f)
    OPQ(c.a10, bits_f32(0x3c8569d7u)) OPQ(c.a8, bits_f32(0x3d4bda59u)) OPQ(c.a6, bits_f32(0x3d886b35u)) OPQ(c.a4, bits_f32(0x3dba2e6eu))
    OPQ(c.a2, bits_f32(0x3e124925u)) OPQ(c.a0t, bits_f32(0x3eaaaaabu))
    OPQ(c.a9, bits_f32(0xbd15a221u)) OPQ(c.a7, bits_f32(0xbd6ef16bu)) OPQ(c.a5, bits_f32(0xbd9d8795u)) OPQ(c.a3, bits_f32(0xbde38e38u)) OPQ(c.a1, bits_f32(0xbe4ccccdu))
    OPQU(c.absmask, 0x7fffffffu) OPQU(c.xlo, 0x38800000u) OPQU(c.tlo, 0x31000000u)
    return c;
}
__device__ __forceinline__ float cheb_sine_scalar_k(float x, const PllConstsP& c) {
    const float z = x * x;
    float p = fmaf(c.c5, z, c.c4);
    p = fmaf(p, z, c.c3);
    p = fmaf(p, z, c.c2);
    p = fmaf(p, z, c.c1);
    p = fmaf(p, z, c.c0);
    return ((z + c.mq25) * x) * p;
}
struct Checks3 { float clamp_max; float half_min; uint32_t r_max; };
__device__ __forceinline__ float pll_step_v3(PllState& s, float p, float q, const PllConstsP& c, Checks3& ck) {
    const float t0 = fmaf(s.lx1, c.b0, s.ly1 * c.a0);
    const float t1 = fmaf(s.err, c.b1, 0.0f);
    const float lpf = (0.0f + t0) + t1;
    s.lx1 = s.err; s.ly1 = lpf;
    const float P = lpf * c.c001;
    s.integ = fmaf(s.err, c.ktsi, s.integ);
    const float PI_error = s.integ + P;
    ck.clamp_max = fmaxf(fmaxf(ck.clamp_max, fabsf(s.integ)), fabsf(PI_error));
    const float freq = fmaf(PI_error, c.m100, c.m19000);
    const float yy = fmaf(freq, c.ts, s.tph);
    s.tph = yy - rintf(yy);
    const float dc = s.tph + c.q25;
    ck.half_min = fminf(fminf(ck.half_min, fabsf(fabsf(yy) - 0.5f)), fabsf(fabsf(dc) - 0.5f));
    const float dt_cos = dc - rintf(dc);
    const float ps = cheb_sine_scalar_k(s.tph, c);
    const float pc = cheb_sine_scalar_k(dt_cos, c);
    const float res_im = fmaf(ps, p, q * pc);
    const float res_re = fmaf(p, pc, -(q * ps));
    const float t = div_unscaled(res_im, res_re);
    // x in [2^-14, 2^13.75) and |t| in [2^-29, 7/16): both windows are 0x0de00000 wide in the exponent/mantissa field
    ck.r_max = max(max(ck.r_max, f32_bits(res_re) - c.xlo), (f32_bits(t) & c.absmask) - c.tlo);
    const float z = t * t;
    const float w = z * z;
    float s1 = c.a8 + w * c.a10;
    s1 = c.a6 + w * s1;
    s1 = c.a4 + w * s1;
    s1 = c.a2 + w * s1;
    s1 = c.a0t + w * s1;
    s1 = z * s1;
    float s2 = c.a7 + w * c.a9;
    s2 = c.a5 + w * s2;
    s2 = c.a3 + w * s2;
    s2 = c.a1 + w * s2;
    s2 = w * s2;
    s.err = t - t * (s1 + s2);
    return s.tph;
}

template <int V>
__global__ __launch_bounds__(kWave) void k_pll_variant(Dims d, const float2* __restrict__ pilot, float* __restrict__ pll_dt,
                                                       float* __restrict__ state, LoopCoeffs k, int power_field,
                                                       unsigned long long* __restrict__ spec_stats) {
    __shared__ __attribute__((aligned(16))) float2 xin[2][kWave * kRowC];
    __shared__ __attribute__((aligned(16))) float dt_out[kWave * kRowF];
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x, c0 = blockIdx.x * kWave, c = c0 + lane;
    const bool live = c < d.C;
    const int cs = live ? c : d.C - 1;
    const int n = d.n_fm_out, chunks = n / kChunk;
    float gain = st(state, S_AGC_PILOT_GAIN, d.C, cs);
    {
        const float sum = st(state, power_field, d.C, cs);
        const float target_gain = sqrtf((1.0f / sum) * (float)n);
        gain = fmaf(target_gain - gain, 0.2f, gain);
    }
    PllState S;
    S.lx1 = st(state, S_PLL_X1, d.C, cs); S.ly1 = st(state, S_PLL_Y1, d.C, cs);
    S.integ = st(state, S_PLL_INT, d.C, cs); S.err = st(state, S_PLL_ERR, d.C, cs); S.tph = st(state, S_PLL_T, d.C, cs);
    int slow_left = 0, hold = 0, n_replayed = 0, n_general = 0;
    const PllConstsP kc = make_consts(k);
    ChunkRegsC regs = chunk_load_c(pilot, n, c0, d.C, 0);
    for (int ch = 0; ch < chunks; ch++) {
        float2* buf = xin[ch & 1];
        chunk_store_c(regs, buf);
        __syncthreads();
        regs = chunk_load_c(pilot, n, c0, d.C, (ch + 1 < chunks ? ch + 1 : ch) * kChunk);
        bool done = false;
        if (slow_left == 0) {
            PllState s = S;
            Checks ck{0.0f, 1.0f, 0u, 0u};
            if (V == 3) {
            } else if (V == 1) {
                for (int t = 0; t < kChunk; t++) {
                    const float2 y = buf[lane * kRowC + t];
                    dt_out[lane * kRowF + t] = pll_step_v1(s, gain * y.x, gain * y.y, k, ck);
                }
            } else {   // V == 2: next sample's LDS read issued one step ahead
                float2 y = buf[lane * kRowC];
#pragma unroll 4
                for (int t = 0; t < kChunk; t++) {
                    const float2 yn = buf[lane * kRowC + (t + 1 < kChunk ? t + 1 : t)];
                    dt_out[lane * kRowF + t] = pll_step_v1(s, gain * y.x, gain * y.y, k, ck);
                    y = yn;
                }
            }
            bool ok = (ck.clamp_max <= 1.0f) && (ck.half_min != 0.0f) && (ck.x_max < 0x3d800000u) && (ck.t_max < 0x0de00000u);
            if (V == 3) {
                s = S;
                Checks3 c3{0.0f, 1.0f, 0u};
                float2 y = buf[lane * kRowC];
#pragma unroll 8
                for (int t = 0; t < kChunk; t++) {
                    const float2 yn = buf[lane * kRowC + (t + 1 < kChunk ? t + 1 : t)];
                    dt_out[lane * kRowF + t] = pll_step_v3(s, gain * y.x, gain * y.y, kc, c3);
                    y = yn;
                }
                ok = (c3.clamp_max <= 1.0f) && (c3.half_min != 0.0f) && (c3.r_max < 0x0de00000u);
            }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) { S = s; done = true; hold = 0; }
            else { slow_left = hold; hold = hold ? (2 * hold < kSlowHoldMax ? 2 * hold : kSlowHoldMax) : 1; n_replayed++; }
        } else {
            slow_left--;
        }
        if (!done) {
            n_general++;
            for (int t = 0; t < kChunk; t++) {
                const float2 y = buf[lane * kRowC + t];
                dt_out[lane * kRowF + t] = pll_step(S, gain * y.x, gain * y.y, k);
            }
        }
        __syncthreads();
        chunk_flush_f(dt_out, pll_dt, n, c0, d.C, ch * kChunk);
    }
    if (live) {
        st(state, S_AGC_PILOT_GAIN, d.C, c) = gain;
        st(state, S_PLL_X1, d.C, c) = S.lx1; st(state, S_PLL_Y1, d.C, c) = S.ly1;
        st(state, S_PLL_INT, d.C, c) = S.integ; st(state, S_PLL_ERR, d.C, c) = S.err; st(state, S_PLL_T, d.C, c) = S.tph;
    }
    if (lane == 0 && spec_stats) {
        atomicAdd(&spec_stats[0], (unsigned long long)chunks);
        atomicAdd(&spec_stats[1], (unsigned long long)n_general);
        atomicAdd(&spec_stats[2], (unsigned long long)n_replayed);
    }
}

// ---- variant 4: two channels per lane, packed fp32 (v_pk_*) ---------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat(float x) { v2f r; r.x = x; r.y = x; return r; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_rint(v2f a) { v2f r; r.x = rintf(a.x); r.y = rintf(a.y); return r; }
__device__ __forceinline__ v2f pk_rcp(v2f a) { v2f r; r.x = __builtin_amdgcn_rcpf(a.x); r.y = __builtin_amdgcn_rcpf(a.y); return r; }
struct PllState2 { v2f lx1, ly1, integ, err, tph; };
struct Checks2 { float tie_min, t_min, t_max, x_min, x_max; };
__device__ __forceinline__ v2f cheb2(v2f x, const PllConsts& c, v2f& zq) {
    const v2f z = x * x;
    v2f p = pk_fma(splat(c.c5), z, splat(c.c4));
    p = pk_fma(p, z, splat(c.c3));
    p = pk_fma(p, z, splat(c.c2));
    p = pk_fma(p, z, splat(c.c1));
    p = pk_fma(p, z, splat(c.c0));
    zq = z + splat(c.mq25);
    return (zq * x) * p;
}
__device__ __forceinline__ v2f pll_step2(PllState2& s, v2f p, v2f q, const PllConsts& c, Checks2& ck) {
    const v2f zero = splat(0.0f);
    const v2f t0 = pk_fma(s.lx1, splat(c.b0), s.ly1 * splat(c.a0));
    const v2f t1 = pk_fma(s.err, splat(c.b1), zero);
    const v2f lpf = (zero + t0) + t1;
    s.lx1 = s.err; s.ly1 = lpf;
    const v2f P = lpf * splat(c.c001);
    s.integ = pk_fma(s.err, splat(c.ktsi), s.integ);
    const v2f PI_error = s.integ + P;
    const v2f freq = pk_fma(PI_error, splat(c.m100), splat(c.m19000));
    const v2f yy = pk_fma(freq, splat(c.ts), s.tph);
    s.tph = yy - pk_rint(yy);
    const v2f dc = s.tph + splat(c.q25);
    const v2f dt_cos = dc - pk_rint(dc);
    v2f zq_s, zq_c;
    const v2f ps = cheb2(s.tph, c, zq_s);
    const v2f pc = cheb2(dt_cos, c, zq_c);
    ck.tie_min = fminf(fminf(ck.tie_min, fabsf(zq_s.x)), fabsf(zq_c.x));
    ck.tie_min = fminf(fminf(ck.tie_min, fabsf(zq_s.y)), fabsf(zq_c.y));
    const v2f res_im = pk_fma(ps, p, q * pc);
    const v2f res_re = pk_fma(p, pc, -(q * ps));
    // div_unscaled
    v2f r = pk_rcp(res_re);
    const v2f nx = -res_re;
    const v2f e0 = pk_fma(nx, r, splat(1.0f));
    r = pk_fma(e0, r, r);
    v2f t = res_im * r;
    const v2f e1 = pk_fma(nx, t, res_im);
    t = pk_fma(e1, r, t);
    const v2f e2 = pk_fma(nx, t, res_im);
    t = pk_fma(e2, r, t);
    ck.t_max = fmaxf(fmaxf(ck.t_max, fabsf(t.x)), fabsf(t.y));
    ck.t_min = fminf(fminf(ck.t_min, fabsf(t.x)), fabsf(t.y));
    ck.x_max = fmaxf(fmaxf(ck.x_max, res_re.x), res_re.y);
    ck.x_min = fminf(fminf(ck.x_min, res_re.x), res_re.y);
    const v2f z = t * t;
    const v2f w = z * z;
    v2f s1 = splat(c.a8) + w * splat(c.a10);
    s1 = splat(c.a6) + w * s1;
    s1 = splat(c.a4) + w * s1;
    s1 = splat(c.a2) + w * s1;
    s1 = splat(c.a0t) + w * s1;
    s1 = z * s1;
    v2f s2 = splat(c.a7) + w * splat(c.a9);
    s2 = splat(c.a5) + w * s2;
    s2 = splat(c.a3) + w * s2;
    s2 = splat(c.a1) + w * s2;
    s2 = w * s2;
    s.err = t - t * (s1 + s2);
    return s.tph;
}

// 16-sample chunk helpers (8 float4 registers per 64-channel group instead of 16)
static constexpr int kCh16 = 16, kRow16C = 18, kRow16F = 20;
struct Chunk16 { float4 v0, v1, v2, v3, v4, v5, v6, v7; };
#define FMD_FOR8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
__device__ __forceinline__ Chunk16 chunk16_load(const float2* __restrict__ base, int n, int c0, int C, int t0) {
    const int lane = threadIdx.x, row = lane >> 3, col = lane & 7;
    Chunk16 r;
#define FMD_LD8(k) { int ch = c0 + 8 * k + row; ch = ch < C ? ch : C - 1; r.v##k = *reinterpret_cast<const float4*>(base + (size_t)ch * n + t0 + 2 * col); }
    FMD_FOR8(FMD_LD8)
#undef FMD_LD8
    return r;
}
__device__ __forceinline__ void chunk16_store(const Chunk16& r, float2* lds) {
    const int lane = threadIdx.x, row = lane >> 3, col = lane & 7;
#define FMD_ST8(k) *reinterpret_cast<float4*>(lds + (8 * k + row) * kRow16C + 2 * col) = r.v##k;
    FMD_FOR8(FMD_ST8)
#undef FMD_ST8
}
__device__ __forceinline__ void chunk16_flush_f(const float* lds, float* __restrict__ out, int n, int c0, int C, int t0) {
    const int lane = threadIdx.x, row = lane >> 2, col = lane & 3;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = 16 * k + row, ch = c0 + r;
        if (ch < C) *reinterpret_cast<float4*>(out + (size_t)ch * n + t0 + 4 * col) = *reinterpret_cast<const float4*>(lds + r * kRow16F + 4 * col);
    }
}

// NP packed pairs per lane: a wavefront owns 128*NP channels; the NP pair streams are independent, so the scheduler
// fills one stream's dependency stalls (and packed-op hazard slots) with the other's instructions
template <int NP>
__global__ __launch_bounds__(kWave) void k_pll_pk(Dims d, const float2* __restrict__ pilot, float* __restrict__ pll_dt,
                                                  float* __restrict__ state, LoopCoeffs k, int power_field,
                                                  unsigned long long* __restrict__ spec_stats) {
    constexpr int G = 2 * NP;   // 64-channel groups per wavefront
    __shared__ __attribute__((aligned(16))) float2 xin[G][kWave * kRow16C];
    __shared__ __attribute__((aligned(16))) float dt_out[G][kWave * kRow16F];
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x, c0 = blockIdx.x * G * kWave;
    const int n = d.n_fm_out, chunks = n / kCh16;
    bool live[G];
    float gain[G];
    PllState S[G];
#pragma unroll
    for (int h = 0; h < G; h++) {
        const int c = c0 + h * kWave + lane;
        live[h] = c < d.C;
        const int cs = live[h] ? c : d.C - 1;
        gain[h] = st(state, S_AGC_PILOT_GAIN, d.C, cs);
        const float sum = st(state, power_field, d.C, cs);
        const float target_gain = sqrtf((1.0f / sum) * (float)n);
        gain[h] = fmaf(target_gain - gain[h], 0.2f, gain[h]);
        S[h].lx1 = st(state, S_PLL_X1, d.C, cs); S[h].ly1 = st(state, S_PLL_Y1, d.C, cs);
        S[h].integ = st(state, S_PLL_INT, d.C, cs); S[h].err = st(state, S_PLL_ERR, d.C, cs); S[h].tph = st(state, S_PLL_T, d.C, cs);
    }
    const PllConsts kc = make_pll_consts(k);
    int slow_left = 0, hold = 0, n_replayed = 0, n_general = 0;
    Chunk16 regs[G];
#pragma unroll
    for (int h = 0; h < G; h++) regs[h] = chunk16_load(pilot, n, c0 + h * kWave, d.C, 0);
    for (int ch = 0; ch < chunks; ch++) {
#pragma unroll
        for (int h = 0; h < G; h++) chunk16_store(regs[h], xin[h]);
        __syncthreads();
        const int tn = (ch + 1 < chunks ? ch + 1 : ch) * kCh16;
#pragma unroll
        for (int h = 0; h < G; h++) regs[h] = chunk16_load(pilot, n, c0 + h * kWave, d.C, tn);
        bool done = false;
        if (slow_left == 0) {
            PllState2 s[NP];
            bool pre = true;
#pragma unroll
            for (int j = 0; j < NP; j++) {
                const PllState &a = S[2 * j], &b = S[2 * j + 1];
                s[j].lx1.x = a.lx1; s[j].lx1.y = b.lx1; s[j].ly1.x = a.ly1; s[j].ly1.y = b.ly1; s[j].integ.x = a.integ; s[j].integ.y = b.integ;
                s[j].err.x = a.err; s[j].err.y = b.err; s[j].tph.x = a.tph; s[j].tph.y = b.tph;
                pre = pre && pll_chunk_precheck(a, k) && pll_chunk_precheck(b, k);
            }
            Checks2 ck{1.0f, 1.0f, 0.0f, 1.0f, 1.0f};
            float2 y[G];
#pragma unroll
            for (int h = 0; h < G; h++) y[h] = xin[h][lane * kRow16C];
#pragma unroll 4
            for (int t = 0; t < kCh16; t++) {
                const int tn2 = (t + 1 < kCh16 ? t + 1 : t);
                float2 yn[G];
#pragma unroll
                for (int h = 0; h < G; h++) yn[h] = xin[h][lane * kRow16C + tn2];
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    v2f p, q;
                    p.x = gain[2 * j] * y[2 * j].x; p.y = gain[2 * j + 1] * y[2 * j + 1].x;
                    q.x = gain[2 * j] * y[2 * j].y; q.y = gain[2 * j + 1] * y[2 * j + 1].y;
                    const v2f dt = pll_step2(s[j], p, q, kc, ck);
                    dt_out[2 * j][lane * kRow16F + t] = dt.x; dt_out[2 * j + 1][lane * kRow16F + t] = dt.y;
                }
#pragma unroll
                for (int h = 0; h < G; h++) y[h] = yn[h];
            }
            bool ok = pre && (ck.tie_min != 0.0f) && (ck.t_max < 0.4375f) && (ck.t_min >= bits_f32(0x31000000u)) &&
                      (ck.x_min >= bits_f32(0x38800000u)) && (ck.x_max < 8192.0f);
#pragma unroll
            for (int j = 0; j < NP; j++) ok = ok && (fabsf(s[j].integ.x) <= 1.0f) && (fabsf(s[j].integ.y) <= 1.0f);
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    PllState &a = S[2 * j], &b = S[2 * j + 1];
                    a.lx1 = s[j].lx1.x; b.lx1 = s[j].lx1.y; a.ly1 = s[j].ly1.x; b.ly1 = s[j].ly1.y; a.integ = s[j].integ.x; b.integ = s[j].integ.y;
                    a.err = s[j].err.x; b.err = s[j].err.y; a.tph = s[j].tph.x; b.tph = s[j].tph.y;
                }
                done = true; hold = 0;
            } else { slow_left = hold; hold = hold ? (2 * hold < kSlowHoldMax ? 2 * hold : kSlowHoldMax) : 1; n_replayed++; }
        } else {
            slow_left--;
        }
        if (!done) {
            n_general++;
            for (int t = 0; t < kCh16; t++) {
#pragma unroll
                for (int h = 0; h < G; h++) {
                    const float2 yy = xin[h][lane * kRow16C + t];
                    dt_out[h][lane * kRow16F + t] = pll_step(S[h], gain[h] * yy.x, gain[h] * yy.y, k);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < G; h++) chunk16_flush_f(dt_out[h], pll_dt, n, c0 + h * kWave, d.C, ch * kCh16);
    }
#pragma unroll
    for (int h = 0; h < G; h++) {
        if (!live[h]) continue;
        const int c = c0 + h * kWave + lane;
        st(state, S_AGC_PILOT_GAIN, d.C, c) = gain[h];
        st(state, S_PLL_X1, d.C, c) = S[h].lx1; st(state, S_PLL_Y1, d.C, c) = S[h].ly1;
        st(state, S_PLL_INT, d.C, c) = S[h].integ; st(state, S_PLL_ERR, d.C, c) = S[h].err; st(state, S_PLL_T, d.C, c) = S[h].tph;
    }
    if (lane == 0 && spec_stats) {
        atomicAdd(&spec_stats[0], (unsigned long long)(G * chunks));
        atomicAdd(&spec_stats[1], (unsigned long long)(G * n_general));
        atomicAdd(&spec_stats[2], (unsigned long long)(G * n_replayed));
    }
}

template <int G>
__global__ __launch_bounds__(kWave) void k_pll_multi(Dims d, const float2* __restrict__ pilot, float* __restrict__ pll_dt,
                                                  float* __restrict__ state, LoopCoeffs k, int power_field,
                                                  unsigned long long* __restrict__ spec_stats) {
    __shared__ __attribute__((aligned(16))) float2 xin[G][kWave * kRow16C];
    __shared__ __attribute__((aligned(16))) float dt_out[G][kWave * kRow16F];
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x, c0 = blockIdx.x * G * kWave;
    const int n = d.n_fm_out, chunks = n / kCh16;
    bool live[G];
    float gain[G];
    PllState S[G];
#pragma unroll
    for (int h = 0; h < G; h++) {
        const int c = c0 + h * kWave + lane;
        live[h] = c < d.C;
        const int cs = live[h] ? c : d.C - 1;
        gain[h] = st(state, S_AGC_PILOT_GAIN, d.C, cs);
        const float sum = st(state, power_field, d.C, cs);
        const float target_gain = sqrtf((1.0f / sum) * (float)n);
        gain[h] = fmaf(target_gain - gain[h], 0.2f, gain[h]);
        S[h].lx1 = st(state, S_PLL_X1, d.C, cs); S[h].ly1 = st(state, S_PLL_Y1, d.C, cs);
        S[h].integ = st(state, S_PLL_INT, d.C, cs); S[h].err = st(state, S_PLL_ERR, d.C, cs); S[h].tph = st(state, S_PLL_T, d.C, cs);
    }
    const PllConsts kc = make_pll_consts(k);
    int slow_left = 0, hold = 0, n_replayed = 0, n_general = 0;
    Chunk16 regs[G];
#pragma unroll
    for (int h = 0; h < G; h++) regs[h] = chunk16_load(pilot, n, c0 + h * kWave, d.C, 0);
    for (int ch = 0; ch < chunks; ch++) {
#pragma unroll
        for (int h = 0; h < G; h++) chunk16_store(regs[h], xin[h]);
        __syncthreads();
        const int tn = (ch + 1 < chunks ? ch + 1 : ch) * kCh16;
#pragma unroll
        for (int h = 0; h < G; h++) regs[h] = chunk16_load(pilot, n, c0 + h * kWave, d.C, tn);
        bool done = false;
        if (slow_left == 0) {
            PllState s[G];
            bool pre = true;
#pragma unroll
            for (int h = 0; h < G; h++) { s[h] = S[h]; pre = pre && pll_chunk_precheck(S[h], k); }
            PllChecks ck{1.0f, 0u};
            float2 y[G];
#pragma unroll
            for (int h = 0; h < G; h++) y[h] = xin[h][lane * kRow16C];
#pragma unroll 4
            for (int t = 0; t < kCh16; t++) {
                const int tn2 = (t + 1 < kCh16 ? t + 1 : t);
                float2 yn[G];
#pragma unroll
                for (int h = 0; h < G; h++) yn[h] = xin[h][lane * kRow16C + tn2];
#pragma unroll
                for (int h = 0; h < G; h++) dt_out[h][lane * kRow16F + t] = pll_step_locked(s[h], gain[h] * y[h].x, gain[h] * y[h].y, kc, ck);
#pragma unroll
                for (int h = 0; h < G; h++) y[h] = yn[h];
            }
            const bool ok = pre && (ck.tie_min != 0.0f) && (ck.range_max < kRangeWindow);
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
                for (int h = 0; h < G; h++) S[h] = s[h];
                done = true; hold = 0;
            } else { slow_left = hold; hold = hold ? (2 * hold < kSlowHoldMax ? 2 * hold : kSlowHoldMax) : 1; n_replayed++; }
        } else {
            slow_left--;
        }
        if (!done) {
            n_general++;
            for (int t = 0; t < kCh16; t++) {
#pragma unroll
                for (int h = 0; h < G; h++) {
                    const float2 yy = xin[h][lane * kRow16C + t];
                    dt_out[h][lane * kRow16F + t] = pll_step(S[h], gain[h] * yy.x, gain[h] * yy.y, k);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < G; h++) chunk16_flush_f(dt_out[h], pll_dt, n, c0 + h * kWave, d.C, ch * kCh16);
    }
#pragma unroll
    for (int h = 0; h < G; h++) {
        if (!live[h]) continue;
        const int c = c0 + h * kWave + lane;
        st(state, S_AGC_PILOT_GAIN, d.C, c) = gain[h];
        st(state, S_PLL_X1, d.C, c) = S[h].lx1; st(state, S_PLL_Y1, d.C, c) = S[h].ly1;
        st(state, S_PLL_INT, d.C, c) = S[h].integ; st(state, S_PLL_ERR, d.C, c) = S[h].err; st(state, S_PLL_T, d.C, c) = S[h].tph;
    }
    if (lane == 0 && spec_stats) {
        atomicAdd(&spec_stats[0], (unsigned long long)(G * chunks));
        atomicAdd(&spec_stats[1], (unsigned long long)(G * n_general));
        atomicAdd(&spec_stats[2], (unsigned long long)(G * n_replayed));
    }
}

__global__ void k_spin_valu(float* out, int iters) {   // dense independent FMAs, no memory
    float a = threadIdx.x * 1e-3f, b = 1.0f, c = 2.0f, e = 3.0f;
    for (int i = 0; i < iters; i++) { a = fmaf(a, 0.999f, 0.1f); b = fmaf(b, 0.999f, 0.1f); c = fmaf(c, 0.999f, 0.1f); e = fmaf(e, 0.999f, 0.1f); }
    if (a + b + c + e == 12345.0f) out[0] = a;
}
__global__ void k_stream_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <typename T> static T* dalloc(size_t n) { T* p; hipMalloc(&p, n * sizeof(T)); hipMemset(p, 0, n * sizeof(T)); return p; }

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 4096;
    const int blocks = argc > 2 ? atoi(argv[2]) : 12;
    Dims d{}; d.C = C; d.m = 1; d.N = 16384; d.n_fm_in = 16384; d.n_fm_out = 8192; d.n_rds = 1024; d.n_audio = 2048; d.n_est = 205;
    const int n = d.n_fm_out;
    // locked pilot, phase-continuous across blocks (8192 * 19000 / 128000 = 1216 whole cycles per block)
    std::vector<float2> hp((size_t)C * n);
    uint32_t lcg = 12345u;
    for (int c = 0; c < C; c++) {
        const double A = 0.02 + 0.001 * (c % 17), ph0 = 0.1 * (c % 61);
        for (int i = 0; i < n; i++) {
            lcg = lcg * 1664525u + 1013904223u; const double n1 = ((lcg >> 8) / 16777216.0 - 0.5) * 0.01 * A;
            lcg = lcg * 1664525u + 1013904223u; const double n2 = ((lcg >> 8) / 16777216.0 - 0.5) * 0.01 * A;
            const double th = 2.0 * M_PI * 19000.0 / 128000.0 * i + ph0;
            hp[(size_t)c * n + i] = make_float2((float)(A * cos(th) + n1), (float)(A * sin(th) + n2));
        }
    }
    float2* pilot = dalloc<float2>((size_t)C * n);
    hipMemcpy(pilot, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
    float* dt[4]; for (auto& p : dt) p = dalloc<float>((size_t)C * n);
    unsigned long long* stats = dalloc<unsigned long long>(8);
    LoopCoeffs k{5.4e-5f, -0.9998f, 1.19f, 0.0024484f, 0.0024484f, 0.9951032f, 0.27f, 0.27f, 0.46f, 0.0019f, 0.0019f, 0.996f};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ref;
    for (int v = 0; v < 4; v++) {
        float* state = dalloc<float>((size_t)S_NUM_FIELDS * C);
        LaunchCtx ctx{}; ctx.d = d; ctx.b.state = state;
        launch_reset_state(ctx, nullptr);
        std::vector<float> pw(C);
        for (int c = 0; c < C; c++) { double s = 0; for (int i = 0; i < n; i++) { const float2 y = hp[(size_t)c * n + i]; s += (double)y.x * y.x + (double)y.y * y.y; } pw[c] = (float)s; }
        hipMemcpy(state + (size_t)S_PILOT_POWER0 * C, pw.data(), C * 4, hipMemcpyHostToDevice);
        hipMemset(stats, 0, 64);
        float last_ms = 0;
        for (int b = 0; b < blocks; b++) {
            if (b == blocks - 3) hipMemset(stats, 0, 64);
            hipEventRecord(e0, nullptr);
            const dim3 g((C + 63) / 64), t(64);
            if (v == 0) hipLaunchKernelGGL(k_pilot_pll, g, dim3(128), 0, nullptr, d, pilot, dt[v], state, k, (int)S_PILOT_POWER0, stats);
            if (v == 1) hipLaunchKernelGGL(k_pll_multi<1>, g, t, 0, nullptr, d, pilot, dt[v], state, k, (int)S_PILOT_POWER0, stats);
            if (v == 2) hipLaunchKernelGGL(k_pll_multi<2>, dim3((C + 127) / 128), t, 0, nullptr, d, pilot, dt[v], state, k, (int)S_PILOT_POWER0, stats);
            if (v == 3) hipLaunchKernelGGL(k_pll_multi<3>, dim3((C + 191) / 192), t, 0, nullptr, d, pilot, dt[v], state, k, (int)S_PILOT_POWER0, stats);
            hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
            hipEventElapsedTime(&last_ms, e0, e1);
            if (b == 0 || b >= blocks - 2) printf("variant %d block %2d: %.3f ms\n", v, b, last_ms);
        }
        unsigned long long hs[8]; hipMemcpy(hs, stats, 64, hipMemcpyDeviceToHost);
        std::vector<float> out((size_t)C * n); hipMemcpy(out.data(), dt[v], out.size() * 4, hipMemcpyDeviceToHost);
        if (v == 0) ref = out;
        size_t diff = 0; for (size_t i = 0; i < out.size(); i++) diff += memcmp(&out[i], &ref[i], 4) != 0;
        printf("variant %d: last-3-block chunks %llu general %llu replayed %llu; dt mismatches vs variant 0: %zu\n", v, hs[0], hs[1], hs[2], diff);
        hipFree(state);
    }
    // the production kernel beside (a) nothing (b) a dense-VALU kernel (c) an HBM streaming copy
    {
        float* state = dalloc<float>((size_t)S_NUM_FIELDS * C);
        LaunchCtx ctx{}; ctx.d = d; ctx.b.state = state;
        launch_reset_state(ctx, nullptr);
        std::vector<float> pw(C);
        for (int c = 0; c < C; c++) { double s2 = 0; for (int i = 0; i < n; i++) { const float2 y = hp[(size_t)c * n + i]; s2 += (double)y.x * y.x + (double)y.y * y.y; } pw[c] = (float)s2; }
        hipMemcpy(state + (size_t)S_PILOT_POWER0 * C, pw.data(), C * 4, hipMemcpyHostToDevice);
        for (int b = 0; b < 10; b++) hipLaunchKernelGGL(k_pilot_pll, dim3((C + 63) / 64), dim3(128), 0, nullptr, d, pilot, dt[0], state, k, (int)S_PILOT_POWER0, stats);
        hipDeviceSynchronize();
        hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        const size_t nb = (size_t)64 << 20;   // 64 Mi float4 = 1 GiB
        float4* ca = dalloc<float4>(nb); float4* cb = dalloc<float4>(nb);
        const char* names[3] = {"alone", "dense VALU beside", "HBM copy beside"};
        for (int mode = 0; mode < 3; mode++) {
            hipMemset(stats, 0, 64); hipDeviceSynchronize();
            if (mode == 1) for (int i = 0; i < 6; i++) hipLaunchKernelGGL(k_spin_valu, dim3(2048), dim3(256), 0, s2, dt[1], 40000);
            if (mode == 2) for (int i = 0; i < 6; i++) hipLaunchKernelGGL(k_stream_copy, dim3(4096), dim3(256), 0, s2, ca, cb, nb);
            hipEventRecord(e0, s1);
            hipLaunchKernelGGL(k_pilot_pll, dim3((C + 63) / 64), dim3(128), 0, s1, d, pilot, dt[0], state, k, (int)S_PILOT_POWER0, stats);
            hipEventRecord(e1, s1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipDeviceSynchronize();
            unsigned long long hs[8]; hipMemcpy(hs, stats, 64, hipMemcpyDeviceToHost);
            printf("k_pilot_pll %-18s: %.3f ms, %llu cycles, clock %.0f MHz, general chunks %llu; phases store %llu compute %llu flush %llu\n", names[mode], ms, hs[6], (double)hs[6] / (double)hs[7] * 100.0, hs[1], hs[3], hs[4], hs[5]);
        }
    }
    return 0;
}
