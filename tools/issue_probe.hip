// Single-wavefront issue/latency probe (development tool): how many cycles per VALU instruction does ONE wave get,
// for 1, 2, 4 independent dependency chains, and for a few instruction kinds.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void k_fma(float* out, long long* cyc, int iters, float a, float b) {
    float x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 0.001f + c;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fmaf(x[c], a, b);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int c = 0; c < CHAINS; c++) s += x[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_div(float* out, long long* cyc, int iters, float a) {
    float x = threadIdx.x * 0.001f + 1.5f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) x = a / x + 1.0f;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_cnd(float* out, long long* cyc, int iters, float a) {
    float x = threadIdx.x * 0.001f + 1.5f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) x = (x > a) ? x * 0.5f : x + 1.0f;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    long long h; const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
#define RUN(name, launch, n_instr) { launch; hipDeviceSynchronize(); hipEventRecord(e0); launch; hipEventRecord(e1); hipEventSynchronize(e1); \
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-28s %7.2f counter-ticks/instr  %7.2f ns/instr\n", name, (double)h / (n_instr), ms * 1e6 / (n_instr)); }
    RUN("fma 1 chain", (k_fma<1><<<1, 64>>>(out, cyc, iters, 0.999f, 0.001f)), (double)iters * 16 * 1);
    RUN("fma 2 chains", (k_fma<2><<<1, 64>>>(out, cyc, iters, 0.999f, 0.001f)), (double)iters * 16 * 2);
    RUN("fma 4 chains", (k_fma<4><<<1, 64>>>(out, cyc, iters, 0.999f, 0.001f)), (double)iters * 16 * 4);
    RUN("fma 8 chains", (k_fma<8><<<1, 64>>>(out, cyc, iters, 0.999f, 0.001f)), (double)iters * 16 * 8);
    RUN("ieee div+add (per pair)", (k_div<<<1, 64>>>(out, cyc, iters, 3.7f)), (double)iters * 4);
    RUN("cmp+cndmask chain (per sel)", (k_cnd<<<1, 64>>>(out, cyc, iters, 2.0f)), (double)iters * 16);
    return 0;
}
