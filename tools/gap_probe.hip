// Development tool: cost of what sits between two kernels of one stream (event record / wait on a completed event / nothing).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
#include <cstdlib>
__global__ void k_busy(float* out, int iters) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, 0.999f, 0.1f);
    if (a == 123.0f) out[0] = a;
}
// fat variant: ~52 KB LDS and >200 VGPRs per wave, two waves (the pilot PLL's footprint)
__global__ __launch_bounds__(128) void k_busy_fat(float* out, int iters) {
    __shared__ float lds[13312];
    float r[200];
#pragma unroll
    for (int i = 0; i < 200; i++) r[i] = threadIdx.x * 0.001f + i;
    lds[threadIdx.x] = r[7];
    __syncthreads();
    float a = lds[(threadIdx.x + 1) & 127];
    for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, 0.999f, 0.1f);
    float acc = a;
#pragma unroll
    for (int i = 0; i < 200; i++) acc += r[i] * a;
    if (acc == 123.0f) out[0] = acc;
}
// background kernel with a FIR-stage footprint: 256 threads, 22 KB LDS
__global__ __launch_bounds__(256) void k_bg(float* out, int iters) {
    __shared__ float lds[5632];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a = lds[(threadIdx.x + 1) & 255], b = a + 1.f, c = a + 2.f, e = a + 3.f;
    for (int i = 0; i < iters; i++) { a = __builtin_fmaf(a, 0.999f, 0.1f); b = __builtin_fmaf(b, 0.999f, 0.1f); c = __builtin_fmaf(c, 0.999f, 0.1f); e = __builtin_fmaf(e, 0.999f, 0.1f); }
    if (a + b + c + e == 123.0f) out[0] = a;
}
int main() {
    float* out; hipMalloc(&out, 4);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ev[64], done; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming); hipEventCreateWithFlags(&done, hipEventDisableTiming);
    const int N = 40, iters = 12000;   // ~160 us kernels
    hipStream_t s3; hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
    const bool bg = getenv("GAP_BG") != nullptr;
    hipEventRecord(done, s2); hipStreamSynchronize(s2);
    hipStream_t s4; hipStreamCreateWithFlags(&s4, hipStreamNonBlocking);
    hipEvent_t evb[64]; for (auto& e : evb) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int mode = 0; mode < 9; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            if (bg) for (int i = 0; i < 30; i++) hipLaunchKernelGGL(k_bg, dim3(65536), dim3(256), 0, s3, out, 400);   // big-grid kernels beside
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (mode == 2 || mode == 3) hipStreamWaitEvent(s1, done, 0);           // dependency already satisfied
                if (mode == 4) { hipEventRecord(ev[(i + 32) % 64], s2); hipStreamWaitEvent(s1, ev[(i + 32) % 64], 0); }   // fresh event on an idle stream
                if (mode >= 6) { hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, s2, out, iters / 4); hipEventRecord(ev[i % 64], s2); hipStreamWaitEvent(s1, ev[i % 64], 0); }   // producer on another stream, done long before
                if (mode == 5) hipExtLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, s1, nullptr, ev[i % 64], 0, out, iters);
                else if (mode == 8) hipLaunchKernelGGL(k_busy_fat, dim3(64), dim3(128), 0, s1, out, iters);
                else hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, s1, out, iters);
                if (mode == 1 || mode == 3 || mode == 4) hipEventRecord(ev[i % 64], s1);
                if (mode == 7) { hipEventRecord(evb[i % 64], s1); hipStreamWaitEvent(s4, evb[i % 64], 0); hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, s4, out, iters / 4); }   // consumer on a third stream
            }
            hipStreamSynchronize(s1); hipStreamSynchronize(s4);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            const char* names[] = {"kernels only", "kernel + record", "wait(done) + kernel", "wait(done) + kernel + record", "wait(fresh other-stream event) + kernel + record", "hipExtLaunch with stop event", "producer stream -> wait + kernel", "producer -> wait + kernel -> record -> consumer stream", "FAT kernels only (52 KB LDS, 200+ VGPRs, 2 waves)"};
            if (rep) printf("%-52s %.1f us per iteration\n", names[mode], us / N);
        }
    }
    return 0;
}
