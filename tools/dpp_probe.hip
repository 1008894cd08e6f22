// What do the gfx9 DPP controls used by k_pll_fast do on this chip?  (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x138, 0xf, 0xf, true);          // wave_shr:1, bound_ctrl (0 for invalid)
    out[64 + l] = __builtin_amdgcn_update_dpp(-7, l, 0x142, 0xa, 0xf, false);    // row_bcast:15 rows 1,3; others keep old (-7)
    out[128 + l] = __builtin_amdgcn_update_dpp(-7, l, 0x111, 0xf, 0xf, true);    // row_shr:1 bound_ctrl
    out[192 + l] = __builtin_amdgcn_update_dpp(-7, l, 0x143, 0xc, 0xf, false);   // row_bcast:31 rows 2,3
}
int main() {
    int* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"wave_shr:1", "row_bcast:15 mask 0xa", "row_shr:1", "row_bcast:31 mask 0xc"};
    for (int t = 0; t < 4; t++) { printf("%s:", names[t]); for (int l = 0; l < 64; l++) printf(" %d", h[64 * t + l]); printf("\n"); }
    return 0;
}
