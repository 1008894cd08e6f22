// Micro-benchmark harness for k_front variants (development tool, not part of the product).
#include "../fm-radio_amd/csrc/fmd_kernels.hip"
#include <cstdio>
#include <vector>
using namespace fmd;

int main(int argc, char** argv) {
    int C = argc > 1 ? atoi(argv[1]) : 4096;
    int m = argc > 2 ? atoi(argv[2]) : 1;
    Dims d{}; d.C = C; d.m = m; d.N = 16384 * m; d.n_fm_in = 16384; d.n_fm_out = 8192; d.n_rds = 1024; d.n_audio = 2048; d.n_est = 205;
    d.tail_base = front_tail_len(m);
    LaunchCtx ctx{}; ctx.d = d;
    size_t nin = (size_t)C * d.N;
    float2* in; hipMalloc(&in, nin * 8);
    std::vector<float2> h(nin);
    for (size_t i = 0; i < nin; i++) { h[i].x = 100.f * cosf(0.001f * (i % 100000)) + (i % 7); h[i].y = 100.f * sinf(0.001f * (i % 100000)) - (i % 5); }
    hipMemcpy(in, h.data(), nin * 8, hipMemcpyHostToDevice);
    for (int p = 0; p < 2; p++) {
        hipMalloc(&ctx.b.base_tail[p], (size_t)C * d.tail_base * 8); hipMemset(ctx.b.base_tail[p], 0, (size_t)C * d.tail_base * 8);
        hipMalloc(&ctx.b.fm_out_iq[p], (size_t)C * d.n_fm_out * 8);
        hipMalloc(&ctx.b.fm_out[p], (size_t)C * d.n_fm_out * 4);
        hipMalloc(&ctx.b.fo_tail[p], (size_t)C * 64 * 4);
    }
    for (int i = 0; i < 64; i++) { ctx.front.b_fm_in[i] = 0.01f * (i + 1); ctx.front.b_fm_out[i] = 0.02f * (64 - i); }
    for (int i = 0; i < 32; i++) ctx.front.b_hilbert_odd[i] = 0.03f * (i - 16);
    ctx.front.fm_gain = 0.27f;
    prepare_kernels();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) launch_stage_front(ctx, it & 1, in, false, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    const int iters = 10;
    for (int it = 0; it < iters; it++) launch_stage_front(ctx, it & 1, in, false, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k_front C=%d m=%d: %.3f ms per launch\n", C, m, ms / iters);
    return 0;
}
