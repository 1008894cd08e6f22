// Development probe (round 6): what a streaming kernel can move through HBM on MI355X.
// MI355X_MICROARCH.md records 6.29 TB/s for a float4 copy; tools/cumask_probe.hip's grid-stride copy (one 16-byte load in flight per lane,
// 8 workgroups per CU) reached 4.6-5.2 and DESIGN.md called that "the chip's practical roof".  This probe varies what that copy left fixed:
// loads in flight per lane (U), workgroups per CU, default / non-temporal policy on either side, and read-only / write-only streams,
// on buffers far beyond the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_roof_probe.hip -o /tmp/hbm_roof_probe && /tmp/hbm_roof_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// every workgroup streams a contiguous chunk; U independent 16-byte loads per lane are issued before the first store
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ in, f4* __restrict__ out, size_t n4) {
    const size_t per = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * per; base < n4; base += (size_t)gridDim.x * per) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            v[u] = NTL ? __builtin_nontemporal_load(in + i) : in[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (NTS) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u];
        }
    }
}
template <int U, bool NTL>
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ in, float* __restrict__ sink, size_t n4) {
    const size_t per = (size_t)256 * U;
    f4 a = {0.f, 0.f, 0.f, 0.f};
    for (size_t base = (size_t)blockIdx.x * per; base < n4; base += (size_t)gridDim.x * per) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const size_t i = base + (size_t)u * 256 + threadIdx.x; v[u] = NTL ? __builtin_nontemporal_load(in + i) : in[i]; }
#pragma unroll
        for (int u = 0; u < U; u++) a += v[u];
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) sink[blockIdx.x] = a.x;      // (never true: keeps the loads)
}
template <int U, bool NTS>
__global__ __launch_bounds__(256) void k_write(f4* __restrict__ out, size_t n4) {
    const size_t per = (size_t)256 * U;
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t base = (size_t)blockIdx.x * per; base < n4; base += (size_t)gridDim.x * per) {
#pragma unroll
        for (int u = 0; u < U; u++) { const size_t i = base + (size_t)u * 256 + threadIdx.x; if (NTS) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
    }
}

static hipEvent_t e0, e1;
template <typename F>
static double timed(F launch, int reps = 5) {
    launch();
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const size_t n4 = (size_t)1 << 26;   // 1 GiB per buffer
    f4 *in, *out; float* sink;
    CK(hipMalloc(&in, n4 * 16)); CK(hipMalloc(&out, n4 * 16)); CK(hipMalloc(&sink, 1 << 20));
    CK(hipMemset(in, 1, n4 * 16)); CK(hipMemset(out, 0, n4 * 16));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = n4 * 16 / 1e9;
#define COPY(U, NTL, NTS, WPC) do { const int g = 256 * WPC; const double ms = timed([&] { hipLaunchKernelGGL((k_copy<U, NTL, NTS>), dim3(g), dim3(256), 0, 0, in, out, n4); }); \
        printf("copy  U=%d ntl=%d nts=%d wg/CU=%2d : %.3f ms  %.2f TB/s (read + write)\n", U, NTL, NTS, WPC, ms, 2 * gb / ms); } while (0)
#define READ(U, NTL, WPC) do { const int g = 256 * WPC; const double ms = timed([&] { hipLaunchKernelGGL((k_read<U, NTL>), dim3(g), dim3(256), 0, 0, in, sink, n4); }); \
        printf("read  U=%d ntl=%d       wg/CU=%2d : %.3f ms  %.2f TB/s\n", U, NTL, WPC, ms, gb / ms); } while (0)
#define WRITE(U, NTS, WPC) do { const int g = 256 * WPC; const double ms = timed([&] { hipLaunchKernelGGL((k_write<U, NTS>), dim3(g), dim3(256), 0, 0, out, n4); }); \
        printf("write U=%d       nts=%d wg/CU=%2d : %.3f ms  %.2f TB/s\n", U, NTS, WPC, ms, gb / ms); } while (0)
    COPY(1, false, false, 8); COPY(1, false, false, 16); COPY(1, false, false, 32);
    COPY(2, false, false, 8); COPY(4, false, false, 4); COPY(4, false, false, 8); COPY(4, false, false, 16); COPY(8, false, false, 4); COPY(8, false, false, 8);
    COPY(4, true, false, 8); COPY(4, false, true, 8); COPY(4, true, true, 8); COPY(4, true, true, 16); COPY(8, true, true, 8); COPY(8, true, true, 4); COPY(1, true, true, 8);
    READ(1, false, 8); READ(4, false, 8); READ(4, true, 8); READ(8, false, 8); READ(8, true, 8); READ(8, true, 4); READ(8, true, 16);
    WRITE(1, false, 8); WRITE(4, false, 8); WRITE(4, true, 8); WRITE(8, true, 8);
    // 8 : 1 read : write, the demodulator's own ratio at 256 kSa/s (8 B in, ~1 B out per sample)
    {
        const size_t nw = n4 / 8;
        const int g = 256 * 8;
        const double ms = timed([&] { hipLaunchKernelGGL((k_read<4, true>), dim3(g), dim3(256), 0, 0, in, sink, n4); hipLaunchKernelGGL((k_write<4, true>), dim3(g), dim3(256), 0, 0, out, nw); });
        printf("read 1 GiB then write 1/8 GiB (two launches, nt): %.3f ms  %.2f TB/s\n", ms, (gb + gb / 8) / ms);
    }
    return 0;
}
