// Development probe (round 5): hipExtStreamCreateWithCUMask on MI355X (8 XCDs x 32 CUs).
//   1. which (XCC, SE, CU) a stream masked with the first K bits / with a bit pattern really runs on — how mask bits map to XCDs;
//   2. whether two streams with complementary masks run side by side (two spin kernels that each fill "their" CUs);
//   3. HBM streaming rate of a copy kernel on K-bit masks (does a part of the chip saturate HBM?).
//   hipcc --offload-arch=gfx950 -O3 tools/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_where(unsigned* out, unsigned long long ticks) {
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = (id & 0xffffu) | ((xcc & 0xfu) << 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// VALU-bound kernel: every wavefront a chain of dependent FMAs
__global__ __launch_bounds__(256) void k_valu(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.3f, e = 0.7f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) { a = fmaf(a, b, 0.5f); c = fmaf(c, b, 0.25f); e = fmaf(e, b, 0.125f); }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a + c + e;
}

static hipStream_t masked(const std::vector<uint32_t>& m) { hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data())); return s; }
static std::vector<uint32_t> first_bits(int k) { std::vector<uint32_t> m(8, 0u); for (int i = 0; i < k; i++) m[i >> 5] |= 1u << (i & 31); return m; }
static std::vector<uint32_t> inv(const std::vector<uint32_t>& a) { auto m = a; for (auto& w : m) w = ~w; return m; }

static void where(hipStream_t s, const char* name, unsigned* d) {
    const int G = 4096;
    hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, s, d, 2000ull);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(G); CK(hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (unsigned v : h) per_xcc[v >> 16].insert((v >> 8) & 0xffu);          // (se, sh, cu) bits 8..15
    printf("%-28s", name);
    int tot = 0;
    for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total CUs %d\n", tot);
}

int main() {
    unsigned* d; CK(hipMalloc(&d, 4096 * 4));
    hipStream_t plain; CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    where(plain, "unmasked", d);
    for (int k : {8, 16, 32, 64, 128, 192}) { char nm[64]; snprintf(nm, 64, "first %d bits", k); hipStream_t s = masked(first_bits(k)); where(s, nm, d); CK(hipStreamDestroy(s)); }
    { std::vector<uint32_t> m(8, 0x55555555u); hipStream_t s = masked(m); where(s, "even bits", d); CK(hipStreamDestroy(s)); }
    { std::vector<uint32_t> m(8, 0x00ff00ffu); hipStream_t s = masked(m); where(s, "bits 0-7 of every 16", d); CK(hipStreamDestroy(s)); }

    // 2. do complementary masks overlap?  VALU-bound kernels with enough workgroups for the whole chip
    float* o; CK(hipMalloc(&o, (size_t)8192 * 256 * 4));
    auto timed = [&](hipStream_t a, hipStream_t b, int ga, int gb, int iters) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        if (ga) hipLaunchKernelGGL(k_valu, dim3(ga), dim3(256), 0, a, o, iters);
        if (gb) hipLaunchKernelGGL(k_valu, dim3(gb), dim3(256), 0, b, o + (size_t)4096 * 256, iters);
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    for (int k : {64, 128, 192}) {
        auto ma = first_bits(k); hipStream_t a = masked(ma), b = masked(inv(ma));
        timed(a, b, 2048, 2048, 100);
        const double ta = timed(a, b, 4096, 0, 2000), tb = timed(a, b, 0, 4096, 2000), tab = timed(a, b, 4096, 4096, 2000);
        const double tp = timed(plain, plain, 4096, 0, 2000);
        printf("valu kernels, split %d/%d: A alone %.3f ms, B alone %.3f ms, both %.3f ms (unmasked alone %.3f)\n", k, 256 - k, ta, tb, tab, tp);
        CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
    }
    // 3. HBM rate of a streaming copy on a part of the chip
    const size_t n4 = (size_t)1 << 26;   // 1 GiB in, 1 GiB out
    float4 *in, *out; CK(hipMalloc(&in, n4 * 16)); CK(hipMalloc(&out, n4 * 16)); CK(hipMemset(in, 1, n4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k : {256, 192, 160, 128, 96, 64, 32}) {
        hipStream_t s = k == 256 ? plain : masked(first_bits(k));
        for (int wgs_per_cu : {8}) {
            const int g = k * wgs_per_cu * 4;
            hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, s, in, out, n4);
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, s, in, out, n4);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("copy on %3d CUs: %.3f ms per GiB+GiB  = %.2f TB/s\n", k, ms / 5, 2.0 * n4 * 16 / (ms / 5 * 1e-3) / 1e12);
        }
        if (k != 256) CK(hipStreamDestroy(s));
    }
    return 0;
}
