// Unit check of the 32-lane scans used by k_pll_fast (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
namespace fmd { static constexpr int kWave = 64; static constexpr int kPllChunk = 128, kPllRing = 256, kPilotSeg = 4;
struct Dims { int C, N, m, n_fm_in, n_fm_out, n_rds, n_audio, n_est, tail_base; }; struct LoopCoeffs { float pilot_k, pilot_a0, pilot_a1, pll_b0, pll_b1, pll_a0, t0, t1, t2, b_0, b_1, b_2; };
struct PilotFastTab { float h1[4], h2[4], m[4][4], mlane[33][4], k, a0, a1; };
enum { SA_X1R, SA_X1I, SA_X2R, SA_X2I, SA_Y1R, SA_Y1I, SA_Y2R, SA_Y2I, S_AGC_PILOT_GAIN, S_PLL_X1, S_PLL_Y1, S_PLL_INT, S_PLL_ERR, S_PLL_T };
__device__ inline float& st(float* s, int f, int C, int c) { return s[f * C + c]; }
__device__ inline float bits_f32(unsigned u) { return __uint_as_float(u); } constexpr unsigned kTwoPiBits = 0x40c90fdbu;
__device__ inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ inline float fast_sin_turns(float t) { return __builtin_amdgcn_sinf(t); } __device__ inline float fast_cos_turns(float t) { return __builtin_amdgcn_cosf(t); }
__device__ inline float fast_atan2f(float y, float x) { return atan2f(y, x); }
#include "../fm-radio_amd/csrc/fmd_kernels_fast.inc"
__global__ void k(const float* in, float* out, float a0) {
    const int lane = threadIdx.x, j = lane & 31;
    float a0_2 = a0 * a0, a0_4 = a0_2 * a0_2, a0_8 = a0_4 * a0_4, a0_row = a0;
    for (int i = 0; i < (j & 15); i++) a0_row *= a0;
    float v = in[lane];
    v = fmaf(a0, dpp_f<kRowShr1>(0.f, v, true), v);
    v = fmaf(a0_2, dpp_f<kRowShr2>(0.f, v, true), v);
    v = fmaf(a0_4, dpp_f<kRowShr4>(0.f, v, true), v);
    v = fmaf(a0_8, dpp_f<kRowShr8>(0.f, v, true), v);
    v = fmaf(a0_row, dpp_from_lower_row(v), v);
    out[lane] = v;
    out[64 + lane] = scan32_add(in[lane]);
    float e1 = dpp_f<kWaveShr1>(0.f, in[lane], true); e1 = (j == 0) ? -5.f : e1;
    out[128 + lane] = e1;
    out[192 + lane] = readlane_halves(in[lane], 31, 63, lane >= 32 ? 0xffffffffu : 0u);
}
}
int main() {
    float h[64], o[256]; for (int i = 0; i < 64; i++) h[i] = sinf(0.37f * i) + 0.1f * i;
    float *din, *dout; hipMalloc(&din, 256); hipMalloc(&dout, 1024); hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
    const float a0 = 0.9951f;
    fmd::k<<<1, 64>>>(din, dout, a0); hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    double worst_v = 0, worst_w = 0;
    for (int g = 0; g < 2; g++) { double v = 0, w = 0; for (int j = 0; j < 32; j++) { v = a0 * v + h[32 * g + j]; w += h[32 * g + j];
        worst_v = fmax(worst_v, fabs(v - o[32 * g + j])); worst_w = fmax(worst_w, fabs(w - o[64 + 32 * g + j])); } }
    printf("weighted scan max err %.3g, prefix sum max err %.3g\n", worst_v, worst_w);
    printf("shr1: %g %g %g ... lane32 %g lane33 %g (want -5 h0 h1 ... -5 h32)  h0=%g h1=%g h32=%g\n", o[128], o[129], o[130], o[160], o[161], h[0], h[1], h[32]);
    printf("readlane halves: lane0 %g (want %g) lane40 %g (want %g)\n", o[192], h[31], o[192 + 40], h[63]);
    return 0;
}
