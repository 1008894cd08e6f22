// Dependent-chain latency of individual VALU ops for ONE resident wavefront (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

#define CHAIN_KERNEL(NAME, INIT, STEP)                                                     \
__global__ void NAME(float* out, long long* cyc, int iters, float a, float b) {          \
    INIT;                                                                                 \
    long long t0 = __builtin_readcyclecounter();                                          \
    for (int i = 0; i < iters; i++) {                                                     \
        _Pragma("unroll") for (int u = 0; u < 16; u++) { STEP; }                          \
    }                                                                                     \
    long long t1 = __builtin_readcyclecounter();                                          \
    out[threadIdx.x] = FIN;                                                               \
    if (threadIdx.x == 0) cyc[0] = t1 - t0;                                               \
}
#define FIN x
CHAIN_KERNEL(k_add, float x = threadIdx.x * 0.001f + 1.0f, x = x + a)
CHAIN_KERNEL(k_mul, float x = threadIdx.x * 0.001f + 1.0f, x = x * a)
CHAIN_KERNEL(k_fma, float x = threadIdx.x * 0.001f + 1.0f, x = __builtin_fmaf(x, a, b))
CHAIN_KERNEL(k_rcp, float x = threadIdx.x * 0.001f + 1.5f, x = __builtin_amdgcn_rcpf(x))
CHAIN_KERNEL(k_trunc, float x = threadIdx.x * 0.001f + 1.5f, x = __builtin_truncf(x) + a)
CHAIN_KERNEL(k_med3, float x = threadIdx.x * 0.001f + 1.5f, x = __builtin_amdgcn_fmed3f(x, a, b))
CHAIN_KERNEL(k_bfi, float x = threadIdx.x * 0.001f + 1.5f, x = __builtin_copysignf(a, x))
CHAIN_KERNEL(k_sqrt, float x = threadIdx.x * 0.001f + 1.5f, x = __builtin_amdgcn_sqrtf(x))
CHAIN_KERNEL(k_cmpsel, float x = threadIdx.x * 0.001f + 1.5f, x = (x > a) ? b : x)
#undef FIN
#define FIN (x + y0 + y1 + y2 + y3)
#define IL_INIT float x = threadIdx.x * 0.001f + 1.0f, y0 = x + 1.f, y1 = x + 2.f, y2 = x + 3.f, y3 = x + 4.f
// one dependent op + k independent ops (k rotating accumulators), cycles reported per group
CHAIN_KERNEL(k_dep1_ind1, IL_INIT, x = __builtin_fmaf(x, a, b); if (u & 1) y0 = __builtin_fmaf(y0, a, b); else y1 = __builtin_fmaf(y1, a, b))
CHAIN_KERNEL(k_dep1_ind2, IL_INIT, x = __builtin_fmaf(x, a, b); y0 = __builtin_fmaf(y0, a, b); y1 = __builtin_fmaf(y1, a, b))
CHAIN_KERNEL(k_dep1_ind3, IL_INIT, x = __builtin_fmaf(x, a, b); y0 = __builtin_fmaf(y0, a, b); y1 = __builtin_fmaf(y1, a, b); y2 = __builtin_fmaf(y2, a, b))
CHAIN_KERNEL(k_dep1_lit, IL_INIT, x = __builtin_fmaf(x, a, 0.123f + u))
CHAIN_KERNEL(k_2chains, IL_INIT, x = __builtin_fmaf(x, a, b); y0 = __builtin_fmaf(y0, a, b))
CHAIN_KERNEL(k_3chains, IL_INIT, x = __builtin_fmaf(x, a, b); y0 = __builtin_fmaf(y0, a, b); y1 = __builtin_fmaf(y1, a, b))
CHAIN_KERNEL(k_rndne, IL_INIT, x = __builtin_rintf(x) + a)
CHAIN_KERNEL(k_max3, IL_INIT, x = __builtin_fmaxf(__builtin_fmaxf(x, a), b) + a)
#undef FIN
#define FIN (x.x + x.y)
#define PK_INIT v2f x; x.x = threadIdx.x * 0.001f + 1.0f; x.y = 2.0f; v2f va; va.x = a; va.y = a; v2f vb; vb.x = b; vb.y = b
CHAIN_KERNEL(k_pkfma, PK_INIT, x = __builtin_elementwise_fma(x, va, vb))
CHAIN_KERNEL(k_pkmul, PK_INIT, x = x * va)
#undef FIN

int main(int argc, char** argv) {
    const int NT = argc > 1 ? atoi(argv[1]) : 64;
    float* out; long long* cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    long long h; const int iters = 20000;
#define RUN(K, aa, bb) { K<<<1, NT>>>(out, cyc, iters, aa, bb); hipDeviceSynchronize(); K<<<1, NT>>>(out, cyc, iters, aa, bb); hipDeviceSynchronize(); \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-10s %6.2f cycles/op\n", #K, (double)h / ((double)iters * 16)); }
    RUN(k_add, 0.001f, 0.f) RUN(k_mul, 0.9999f, 0.f) RUN(k_fma, 0.999f, 0.001f) RUN(k_rcp, 0.f, 0.f) RUN(k_trunc, 0.37f, 0.f)
    RUN(k_med3, -1.f, 1.f) RUN(k_bfi, 0.5f, 0.f) RUN(k_sqrt, 0.f, 0.f) RUN(k_cmpsel, 2.0f, 1.7f) RUN(k_pkfma, 0.999f, 0.001f) RUN(k_pkmul, 0.9999f, 0.f)
    RUN(k_dep1_ind1, 0.999f, 0.001f) RUN(k_dep1_ind2, 0.999f, 0.001f) RUN(k_dep1_ind3, 0.999f, 0.001f) RUN(k_dep1_lit, 0.999f, 0.001f) RUN(k_2chains, 0.999f, 0.001f) RUN(k_3chains, 0.999f, 0.001f) RUN(k_rndne, 0.3f, 0.f) RUN(k_max3, 0.3f, 0.2f)
    return 0;
}
