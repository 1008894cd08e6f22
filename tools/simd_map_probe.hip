// Development probe: where do the wavefronts of one workgroup land (SIMD id from HW_REG_HW_ID), and do four busy wavefronts of one
// workgroup share VALU issue?   hipcc --offload-arch=gfx950 -O3 tools/simd_map_probe.hip -o /tmp/simd_map_probe && /tmp/simd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_where(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = id;
}
__global__ void k_chain(float* out, int iters, int active_waves) {
    const int w = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    if (w < active_waves) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) a = fmaf(a, b, 0.5f);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 64 * 4 * 4);
    for (int waves : {3, 4}) {
        hipLaunchKernelGGL(k_where, dim3(8), dim3(64 * waves), 0, 0, d);
        std::vector<unsigned> h(8 * waves); (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        for (int b = 0; b < 8; b++) { printf("wg %d (%d waves):", b, waves); for (int w = 0; w < waves; w++) { unsigned id = h[b * waves + w]; printf("  [wave %u simd %u cu %u se %u]", id & 15, (id >> 4) & 3, (id >> 8) & 15, (id >> 13) & 7); } printf("\n"); }
    }
    float* o; (void)hipMalloc(&o, 64 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int act = 1; act <= 4; act++) {
        hipLaunchKernelGGL(k_chain, dim3(64), dim3(256), 0, 0, o, 1000, act);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_chain, dim3(64), dim3(256), 0, 0, o, 20000, act); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("active waves %d of 4: %.3f ms (%.2f ns per dependent fma)\n", act, ms, ms * 1e6 / (20000.0 * 16));
    }
    return 0;
}
