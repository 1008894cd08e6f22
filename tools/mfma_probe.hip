// f32 MFMA as an exact fmaf chain (development tool): operand layout and bit-exactness of v_mfma_f32_4x4x1_16b_f32,
// its issue rate / dependent latency, and how it shares a SIMD with a VALU-bound wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* a, const float* b, const float* c, float* d, int steps) {
    const int l = threadIdx.x;
    v4f acc = {c[l], c[64 + l], c[128 + l], c[192 + l]};
    for (int k = 0; k < steps; k++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k * 64 + l], b[k * 64 + l], acc, 0, 0, 0);
    d[l] = acc[0]; d[64 + l] = acc[1]; d[128 + l] = acc[2]; d[192 + l] = acc[3];
}

// MODE 0: NACC independent accumulators, back to back.  Cycles per MFMA for one wave.
template <int NACC>
__global__ void k_rate(float* out, long long* cyc, int iters, float a, float b) {
    v4f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float av = a + threadIdx.x * 1e-3f, bv = b;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
__global__ void k_rate16(float* out, long long* cyc, int iters, float a, float b) {
    v4f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float av = a + threadIdx.x * 1e-3f, bv = b;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// 8 waves per workgroup = 2 per SIMD.  role bit0: waves 0-3 run MFMAs, bit1: waves 4-7 run VALU fmas.
__global__ __launch_bounds__(512) void k_share(float* out, long long* cyc, int iters, float a, float b, int roles) {
    const int w = threadIdx.x / 64;
    const bool mf = (w < 4);
    float s = 0.f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (mf) {
        if (roles & 1) {
            v4f acc[4];
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
            float av = a + threadIdx.x * 1e-3f;
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, b, acc[i], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else if (roles & 2) {
        float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            }
        }
        s = x0 + x1 + x2 + x3;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[w] = t1 - t0;
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
    // ---- layout + exactness
    const int steps = 35;
    std::vector<float> a(steps * 64), b(steps * 64), c(256), d(256);
    srand(1);
    auto rnd = []() { return (float)((rand() % 200001) - 100000) * 1.1e-5f; };
    for (auto& v : a) v = rnd();
    for (auto& v : b) v = rnd();
    for (auto& v : c) v = rnd();
    float *da, *db, *dc, *dd;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dc, 1024); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(da, db, dc, dd, steps);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane = 4*block + j; A row i of block comes from lane 4*block + i; D vgpr i, lane 4*block + j = row i, column j
    int bad = 0;
    for (int v = 0; v < 4; v++)
        for (int l = 0; l < 64; l++) {
            const int blk = l / 4;
            float acc = c[v * 64 + l];
            for (int k = 0; k < steps; k++) acc = fmaf(a[k * 64 + 4 * blk + v], b[k * 64 + l], acc);
            if (bits(acc) != bits(d[v * 64 + l])) bad++;
        }
    printf("4x4x1_16b: D[vgpr i][lane 4b+j] = fma-chain(A[lane 4b+i], B[lane 4b+j]) over %d steps: %d of 256 differ\n", steps, bad);

    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    long long h[64];
    const int iters = 4000;
#define RATE(K, NACC, NT) { K<NACC><<<1, NT>>>(out, cyc, iters, 0.5f, 0.25f); hipDeviceSynchronize(); K<NACC><<<1, NT>>>(out, cyc, iters, 0.5f, 0.25f); hipDeviceSynchronize(); \
    hipMemcpy(h, cyc, 8 * (NT / 64), hipMemcpyDeviceToHost); printf("%-9s acc=%d waves=%d: %6.2f cycles per MFMA per wave\n", #K, NACC, NT / 64, (double)h[0] / ((double)iters * 8 * NACC)); }
    RATE(k_rate, 1, 64) RATE(k_rate, 2, 64) RATE(k_rate, 4, 64) RATE(k_rate, 8, 64) RATE(k_rate, 4, 256) RATE(k_rate, 4, 512)
    RATE(k_rate16, 1, 64) RATE(k_rate16, 2, 64) RATE(k_rate16, 4, 64) RATE(k_rate16, 4, 512)
    for (int roles = 1; roles <= 3; roles++) {
        k_share<<<1, 512>>>(out, cyc, iters, 0.999f, 0.001f, roles); hipDeviceSynchronize();
        k_share<<<1, 512>>>(out, cyc, iters, 0.999f, 0.001f, roles); hipDeviceSynchronize();
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("share roles=%d: mfma wave %7.2f cycles/MFMA, valu wave %6.2f cycles/fma\n", roles, (double)h[0] / (iters * 32.0), (double)h[4] / (iters * 32.0));
    }
    return 0;
}
