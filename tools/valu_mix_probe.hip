// Chip-level VALU issue rate by operand pattern (development tool): W waves per SIMD, 8 independent chains per wave, every
// instruction spelled in inline asm so that the compiler can neither pack nor re-form it.
// Result on MI355X (cycles per wave64 instruction and SIMD, 4-8 waves per SIMD):  v_fma/v_fmac with VGPR sources 2.75,
// v_mov 2.45, **v_fmac/v_fma with an SGPR source 4.2**, v_pk_fma (VGPR or SGPR sources) ~5 (2.5 per FMA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x[8], y[8], z[8];
    f2v px[8], py[8];
    __shared__ float4 lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(a, b, a, b);
    __syncthreads();
    for (int c = 0; c < 8; c++) { x[c] = threadIdx.x * 0.001f + c; y[c] = x[c] * 0.5f + 1.0f; z[c] = 0.25f + c; px[c] = f2v{x[c], y[c]}; py[c] = f2v{z[c], x[c]}; }
    f2v ab{a, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float4 t = make_float4(0, 0, 0, 0);
            if (KIND == 8) t = lds[(i + u) & 63];   // broadcast read of four "taps"
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y[c]), "v"(z[c]));
                else if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "v"(y[c]), "v"(z[c]));
                else if (KIND == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "s"(a), "v"(z[c]));
                else if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y[c]), "s"(a));
                else if (KIND == 4) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[c]) : "s"(a), "v"(z[c]));
                else if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(px[c]) : "v"(py[c]), "s"(ab));
                else if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(px[c]) : "v"(py[c]), "v"(px[(c + 1) & 7]));
                else if (KIND == 7) asm volatile("v_fmac_f32 %0, 0x3f7fbe77, %1" : "+v"(x[c]) : "v"(z[c]));          // literal
                else if (KIND == 8) { const float tt = (c & 3) == 0 ? t.x : (c & 3) == 1 ? t.y : (c & 3) == 2 ? t.z : t.w;
                                      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "v"(tt), "v"(z[c])); }
                else if (KIND == 9) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[c]) : "v"(y[c]), "v"(z[c]));
                else if (KIND == 10) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[c]) : "v"(y[c]), "v"(z[c]));
                else if (KIND == 11) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[c]) : "v"(y[c]), "v"(z[c]));
                else if (KIND == 12) asm volatile("v_fma_f32 %0, %1, 2.0, %0" : "+v"(x[c]) : "v"(y[c]));             // inline constant
            }
        }
    }
    float s = 0; for (int c = 0; c < 8; c++) s += x[c] + y[c] + z[c] + px[c].x + px[c].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND> void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    const int iters = 2000;
    for (int wps : {2, 4, 8}) {
        const int blocks = 1024 * wps;
        k<KIND><<<blocks, 64>>>(out, iters, 0.999f, 0.001f); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k<KIND><<<blocks, 64>>>(out, iters, 0.999f, 0.001f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * iters * 64;
        printf("%-38s %d waves/SIMD: %8.1f G wave-instr/s = %.2f cycles/instr/SIMD at 2.4 GHz\n", name, wps, instr / ms / 1e6, 1024.0 * 2.4e9 / (instr / (ms * 1e-3)));
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 4 * 64 * 1024 * 16);
    run<0>("v_fma_f32 v,v,v,v", out);
    run<1>("v_fmac_f32 v,v,v", out);
    run<2>("v_fmac_f32 v,s,v", out);
    run<3>("v_fma_f32 v,v,s,v", out);
    run<4>("v_mul_f32 v,s,v", out);
    run<5>("v_pk_fma_f32 v,v,s,v", out);
    run<6>("v_pk_fma_f32 v,v,v,v", out);
    run<7>("v_fmac_f32 v,literal,v", out);
    run<8>("v_fmac_f32 v,v,v + ds_read_b128/8", out);
    run<9>("v_add_f32 v,v,v", out);
    run<10>("v_add_u32 v,v,v", out);
    run<11>("v_cndmask_b32 v,v,v,vcc", out);
    run<12>("v_fma_f32 v,v,2.0,v", out);
    return 0;
}
