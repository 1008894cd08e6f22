// Development probe (GPU box): what a dependent launch costs on this runtime, by mechanism and grid size.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_probe.hip -o tools/launch_probe && tools/launch_probe
// A streaming kernel (grid x 256 threads, ~`us` microseconds of work) is launched N times on one stream with
//   plain       hipLaunchKernelGGL back to back
//   ext_stop    hipExtLaunchKernelGGL carrying a stop event (how libfmdemod attaches a stage's "done" event to its dispatch)
//   record      plain launch + hipEventRecord behind it
//   wait        hipStreamWaitEvent on an event of ANOTHER stream that completed long ago, then a plain launch
//   wait+ext    both (what a pipelined stage's launch looks like)
//   pending     wait+ext, but the awaited event is still PENDING when the host submits (the host runs ahead, as in the pipeline):
//               it is recorded on a second stream behind a wait for this stream's launch three iterations back, so on the device it
//               completes two launches before it is needed — the runtime cannot elide the barrier packet, yet it never really blocks
// and the average time per iteration is printed.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_tiny(int* p) { if (threadIdx.x == 0 && p) p[0] = 1; }
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ in, float4* __restrict__ out, int per_block) {
    const size_t base = (size_t)blockIdx.x * per_block;
    for (int i = threadIdx.x; i < per_block; i += 256) {
        float4 v = in[base + i];
        v.x = v.x * 1.0001f + v.y; v.z = v.z * 0.9999f + v.w;
        out[base + i] = v;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int N = 200;
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(N + 32), e2(N + 32);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : e2) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t old_ev;
    CK(hipEventCreateWithFlags(&old_ev, hipEventDisableTiming));
    const size_t total_f4 = (size_t)32768 * 4096;    // 2 GB in, 2 GB out
    float4 *in, *out;
    CK(hipMalloc(&in, total_f4 * 16)); CK(hipMalloc(&out, total_f4 * 16));
    CK(hipMemset(in, 0, total_f4 * 16));
    for (int grid : {64, 2048, 32768}) {
        const int per_block = (int)(total_f4 / 4 / grid);          // a quarter of the buffer: ~1 GB of traffic per launch
        CK(hipEventRecord(old_ev, s2)); CK(hipStreamSynchronize(s2));
        for (int mode = 0; mode < 6; mode++) {
            auto go = [&](int i) -> hipError_t {
                if (mode == 5) {
                    hipError_t e = hipSuccess;
                    if (i >= 3) e = hipStreamWaitEvent(s2, ev[i - 3], 0);
                    if (e != hipSuccess) return e;
                    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s2, (int*)nullptr);
                    e = hipEventRecord(e2[i], s2); if (e != hipSuccess) return e;
                    e = hipStreamWaitEvent(s, e2[i], 0); if (e != hipSuccess) return e;
                    hipExtLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, nullptr, ev[i], 0, in, out, per_block);
                    return hipGetLastError();
                }
                if (mode == 3 || mode == 4) { hipError_t e = hipStreamWaitEvent(s, old_ev, 0); if (e != hipSuccess) return e; }
                if (mode == 1 || mode == 4) hipExtLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, nullptr, ev[i], 0, in, out, per_block);
                else hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, in, out, per_block);
                if (mode == 2) return hipEventRecord(ev[i], s);
                return hipGetLastError();
            };
            for (int i = 0; i < 20; i++) CK(go(i));
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) CK(go(i));
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            const auto t2 = std::chrono::steady_clock::now();
            const char* names[6] = {"plain", "ext_stop", "record", "wait", "wait+ext", "pending"};
            printf("grid %6d  %-9s  %.1f us per iteration (host submit %.1f us)\n", grid, names[mode],
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        }
    }
    return 0;
}
