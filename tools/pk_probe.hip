// Development probe: VALU issue cost of packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against the scalar forms, and of the
// transcendentals, with W wavefronts per SIMD (a full chip's worth of workgroups), in ns per wave-instruction and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[8]; f2 p[8];
    for (int c = 0; c < 8; c++) { x[c] = threadIdx.x * 0.001f + c; p[c] = f2{x[c], x[c] + 0.5f}; }
    const f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pa), "v"(pb));
                if constexpr (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pa));
                if constexpr (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pb));
                if constexpr (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
                if constexpr (KIND == 5) asm volatile("v_sin_f32 %0, %0" : "+v"(x[c]));
                if constexpr (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a));
                if constexpr (KIND == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if constexpr (KIND == 8) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[c]));
                if constexpr (KIND == 9) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if constexpr (KIND == 10) asm volatile("v_fract_f32 %0, %0" : "+v"(x[c]));
                if constexpr (KIND == 11) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[c]));
                if constexpr (KIND == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x[c]) : "v"(a) : "s10", "s11");
                if constexpr (KIND == 13) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x[c]), "v"(a) : "vcc");
                if constexpr (KIND == 14) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc");
                if constexpr (KIND == 15) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if constexpr (KIND == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if constexpr (KIND == 17) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if constexpr (KIND == 18) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if constexpr (KIND == 19) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if constexpr (KIND == 20) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if constexpr (KIND == 21) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[c]));
                if constexpr (KIND == 22) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if constexpr (KIND == 23) asm volatile("v_cmp_gt_f32 s[10:11], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[10:11]" : "+v"(x[c]) : "v"(a), "v"(b) : "s10", "s11");
                if constexpr (KIND == 24) asm volatile("v_fma_f32 %0, |%0|, %1, -%2" : "+v"(x[c]) : "v"(a), "v"(b));
                if constexpr (KIND == 25) asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(a));
                if constexpr (KIND == 26) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[c]));
                if constexpr (KIND == 27) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x[c]));
            }
        }
    }
    float s = 0; for (int c = 0; c < 8; c++) s += x[c] + p[c].x + p[c].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 4 * 256 * 4096);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f32", "v_sin_f32", "v_cndmask_b32", "v_perm_b32", "v_rndne_f32", "v_max3_f32", "v_fract_f32", "v_mov dpp row_shr", "v_cndmask e64 sgpr", "v_cmp_gt_f32 vcc", "v_cmp+v_cndmask vcc", "v_bfi_b32", "v_and_b32", "v_mul_f32", "v_add_f32", "v_fmac_f32", "v_sub_f32", "v_lshlrev_b32", "v_min_f32", "v_cmp+cndmask sgpr", "v_fma_f32 |x| -c", "v_mov_b32", "v_cvt_f32_i32", "v_cvt_f32_ubyte0"};
    for (int wps = 2; wps <= 8; wps *= 4) {           // wavefronts per SIMD
        const int blocks = 256 * wps;                  // 256 CUs x 4 SIMDs x wps waves = blocks x 4 waves
#define RUN(K) { k<K><<<blocks, 256>>>(out, iters, 0.999f, 0.001f); hipDeviceSynchronize(); hipEventRecord(e0); k<K><<<blocks, 256>>>(out, iters, 0.999f, 0.001f); hipEventRecord(e1); hipEventSynchronize(e1); \
    hipEventElapsedTime(&ms, e0, e1); printf("%d waves/SIMD  %-20s %6.2f ns per wave-instruction and SIMD\n", wps, names[K], ms * 1e6 / ((double)iters * 64 * wps)); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27)
    }
    return 0;
}
