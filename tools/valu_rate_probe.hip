// Chip-level VALU issue-rate probe (development tool): aggregate wave-instructions per second with W waves per SIMD, each wave
// running 8 independent chains, for scalar v_fma_f32, packed v_pk_fma_f32 and a DPP add.  Settles what "100 % VALU issue" is.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x[8]; f2v y[8];
    for (int c = 0; c < 8; c++) { x[c] = threadIdx.x * 0.001f + c; y[c] = f2v{x[c], x[c] + 0.5f}; }
    f2v av{a, a}, bv{b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (KIND == 0) x[c] = __builtin_fmaf(x[c], a, b);
                else if (KIND == 1) y[c] = __builtin_elementwise_fma(y[c], av, bv);
                else if (KIND == 2) x[c] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[c]), 0x111, 0xf, 0xf, true));
                else if (KIND == 3) x[c] = __builtin_amdgcn_sinf(x[c]);
            }
        }
    }
    float s = 0; for (int c = 0; c < 8; c++) s += x[c] + y[c].x + y[c].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 4 * 64 * 1024 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    const int iters = 4000;
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32 dpp", "v_sin_f32"};
    for (int kind = 0; kind < 4; kind++)
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 1024 * wps;
            auto launch = [&]() {
                if (kind == 0) k<0><<<blocks, 64>>>(out, iters, 0.999f, 0.001f);
                else if (kind == 1) k<1><<<blocks, 64>>>(out, iters, 0.999f, 0.001f);
                else if (kind == 2) k<2><<<blocks, 64>>>(out, iters, 0.999f, 0.001f);
                else k<3><<<blocks, 64>>>(out, iters, 0.999f, 0.001f);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks * iters * 64;
            printf("%-14s %d waves/SIMD: %8.1f G wave-instr/s  = %.2f cycles/instr/SIMD at 2.4 GHz\n", names[kind], wps, instr / ms / 1e6, 1024.0 * 2.4e9 / (instr / (ms * 1e-3)));
        }
    return 0;
}
