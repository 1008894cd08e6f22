// Development micro-benchmark (GPU box): what lies between two launches of a serial kernel on one queue — plain back to back, behind an
// event wait on another queue's event that completed long ago, and behind one recorded just before (the RDS stage's situation).
//   hipcc --offload-arch=gfx950 -O2 tools/launch_gap_probe.hip -o /tmp/launch_gap_probe && /tmp/launch_gap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out && threadIdx.x == 0) out[blockIdx.x] = wall_clock64();
}
int main() {
    hipStream_t sR, sF;
    hipStreamCreateWithFlags(&sR, hipStreamNonBlocking); hipStreamCreateWithFlags(&sF, hipStreamNonBlocking);
    const int N = 40;
    std::vector<hipEvent_t> t0(N), t1(N), dep(N);
    for (int i = 0; i < N; i++) { hipEventCreate(&t0[i]); hipEventCreate(&t1[i]); hipEventCreateWithFlags(&dep[i], hipEventDisableTiming); }
    auto run = [&](int mode, const char* name) {
        hipDeviceSynchronize();
        if (mode == 1) { for (int i = 0; i < N; i++) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, sF, 100ull, nullptr); hipEventRecord(dep[i], sF); } hipStreamSynchronize(sF); }
        for (int i = 0; i < N; i++) {
            if (mode == 2) { hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, sF, 3000ull, nullptr); hipEventRecord(dep[i], sF); }      // a 30 us kernel on the other queue, recorded just now
            if (mode >= 1) hipStreamWaitEvent(sR, dep[i], 0);
            hipExtLaunchKernelGGL(spin, dim3(16), dim3(320), 0, sR, t0[i], t1[i], 0, 10000ull, nullptr);                               // 100 us at 100 MHz ticks
        }
        hipDeviceSynchronize();
        double gap = 0, dur = 0;
        for (int i = 1; i < N; i++) { float g = 0, d = 0; hipEventElapsedTime(&g, t1[i - 1], t0[i]); hipEventElapsedTime(&d, t0[i], t1[i]); gap += g; dur += d; }
        std::printf("%-62s kernel %.1f us, between two launches %.1f us\n", name, 1e3 * dur / (N - 1), 1e3 * gap / (N - 1));
    };
    for (int rep = 0; rep < 2; rep++) {
        run(0, "plain, back to back on one queue");
        run(1, "each behind hipStreamWaitEvent on an event complete long ago");
        run(2, "each behind hipStreamWaitEvent on an event recorded just before");
    }
    return 0;
}
