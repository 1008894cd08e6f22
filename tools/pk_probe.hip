// Development probe (GPU box): issue rate of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 against their scalar forms, and of v_sin_f32.
// hipcc --offload-arch=gfx950 -O3 tools/pk_probe.hip -o tools/pk_probe && tools/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                v2f v = {x[i], x[i + 1]}, aa = {a, a}, bb = {b, b};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v) : "v"(aa), "v"(bb));
                x[i] = v[0]; x[i + 1] = v[1];
            }
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_sin_f32 %0, %0" : "+v"(x[i]));
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                v2f v = {x[i], x[i + 1]}, aa = {a, a};
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(v) : "v"(aa));
                x[i] = v[0]; x[i + 1] = v[1];
            }
        } else if constexpr (KIND == 4) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
        } else if constexpr (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        } else if constexpr (KIND == 6) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pack_b32_f16 %0, %0, %1 op_sel:[1,1,0]" : "+v"(x[i]) : "v"(a));
        } else if constexpr (KIND == 7) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(x[i]) : "v"(a));
        } else if constexpr (KIND == 8) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        } else if constexpr (KIND == 9) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        } else if constexpr (KIND == 10) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(0x07060302));
        } else if constexpr (KIND == 11) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "s"(0xffff0000));
        } else if constexpr (KIND == 12) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x[i]) : "v"(a));
        } else if constexpr (KIND == 13) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i]));
        } else if constexpr (KIND == 14) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cos_f32 %0, %0" : "+v"(x[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, int insts_per_iter) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000, grid = 256 * 8;       // 8 workgroups per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<grid, 256>>>(out, 10, 1.0001f, 1e-9f);
    hipEventRecord(e0);
    k<KIND><<<grid, 256>>>(out, iters, 1.0001f, 1e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: grid * 4 waves / (256 CUs * 4 SIMDs) * iters * insts
    const double per_simd = (double)grid * 4 / 1024 * iters * insts_per_iter;
    printf("%-14s %.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 16); run<1>("v_pk_fma_f32", 8); run<4>("v_mul_f32", 16); run<3>("v_pk_mul_f32", 8); run<2>("v_sin_f32", 16); run<14>("v_cos_f32", 16); run<5>("v_perm_b32", 16);
    run<10>("v_perm sgpr sel", 16); run<6>("v_pack_b32_f16", 16); run<7>("v_mov_sdwa", 16); run<8>("v_cvt_pk_bf16", 16); run<9>("v_and_or_b32", 16); run<11>("v_and_b32", 16);
    run<12>("v_alignbit", 16); run<13>("v_rndne_f32", 16);
    return 0;
}
