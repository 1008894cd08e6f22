// Development probe (GPU box): lane mapping of gfx950's v_permlane16_swap / v_permlane32_swap as the builtins expose them.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    int a = 100 + l, b = 200 + l;
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r16[0]; out[64 + l] = r16[1]; out[128 + l] = r32[0]; out[192 + l] = r32[1];
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"swap16.v0", "swap16.v1", "swap32.v0", "swap32.v1"};
    for (int k2 = 0; k2 < 4; k2++) { printf("%s:", nm[k2]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%d", l, h[64 * k2 + l]); printf("\n"); }
    return 0;
}
