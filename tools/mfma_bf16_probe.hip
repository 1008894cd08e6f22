// Operand layout of v_mfma_f32_16x16x32_bf16 on gfx950 (development tool): checks the lane <-> (row, k) mapping the FIR kernels assume
//   A (16 x 32): lane l holds A[l % 16][8 (l / 16) + 0..7]      B (32 x 16): lane l holds B[8 (l / 16) + 0..7][l % 16]
//   D (16 x 16): lane l holds D[4 (l / 16) + i][l % 16], i = 0..3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint4* a, const uint4* b, float4* out) {
    bf16x8 A = __builtin_bit_cast(bf16x8, a[threadIdx.x]), B = __builtin_bit_cast(bf16x8, b[threadIdx.x]);
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c, 0, 0, 0);
    out[threadIdx.x] = make_float4(c[0], c[1], c[2], c[3]);
}
static uint16_t bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
static float fb(uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; }
int main() {
    float A[16][32], B[32][16];
    for (auto& r : A) for (float& v : r) v = fb(bf((float)(rand() % 17 - 8) / 4.0f));
    for (auto& r : B) for (float& v : r) v = fb(bf((float)(rand() % 13 - 6) / 8.0f));
    uint16_t ha[64][8], hb[64][8];
    for (int l = 0; l < 64; l++) for (int i = 0; i < 8; i++) { ha[l][i] = bf(A[l % 16][8 * (l / 16) + i]); hb[l][i] = bf(B[8 * (l / 16) + i][l % 16]); }
    uint4 *da, *db; float4* dout;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, 64 * sizeof(float4));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dout);
    float out[64][4];
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
        const int m = 4 * (l / 16) + i, n = l % 16;
        double ref = 0; for (int kk = 0; kk < 32; kk++) ref += (double)A[m][kk] * B[kk][n];
        worst = fmax(worst, fabs(ref - out[l][i]));
    }
    printf("v_mfma_f32_16x16x32_bf16 layout check: worst |D - A B| = %g (%s)\n", worst, worst < 1e-4 ? "layout as assumed" : "LAYOUT DIFFERS");
    return 0;
}
