// Development probe: which HIP streams share a hardware queue?  Two ~200 us spin kernels on two streams take ~200 us when the streams
// have queues of their own and ~400 us when they share one.  Creates D streams first (and uses them), then 8, and prints the groups.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(unsigned long long ticks) {   // 100 MHz real-time counter
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static double pair_us(hipStream_t a, hipStream_t b) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    spin<<<1, 64, 0, a>>>(20000); spin<<<1, 64, 0, b>>>(20000);
    hipStreamSynchronize(a); hipStreamSynchronize(b);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 0, N = 8;
    std::vector<hipStream_t> dummy(D), s(N);
    for (auto& x : dummy) { hipStreamCreateWithFlags(&x, hipStreamNonBlocking); spin<<<1, 64, 0, x>>>(10); }
    hipDeviceSynchronize();
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    spin<<<1, 64, 0, s[0]>>>(10); hipDeviceSynchronize();
    printf("D=%d  pair times (us), rows/cols = streams in creation order:\n", D);
    for (int i = 0; i < N; i++) { for (int j = 0; j < N; j++) printf("%5.0f", i == j ? 0.0 : pair_us(s[i], s[j])); printf("\n"); }
    printf("with the null stream:"); for (int i = 0; i < N; i++) printf("%5.0f", pair_us(s[i], nullptr)); printf("\n");
    return 0;
}
