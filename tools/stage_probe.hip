// Times each pipeline stage in isolation on synthetic data (development tool, not part of the product).
//   stage_probe [C=4096] [m=1] [iters=10]
#ifndef PROBE_KERNELS
#define PROBE_KERNELS "../fm-radio_amd/csrc/fmd_kernels.hip"
#endif
#include PROBE_KERNELS
#include <cstdio>
#include <vector>
using namespace fmd;

template <typename T> static T* dalloc(size_t n) { T* p; hipMalloc(&p, n * sizeof(T)); hipMemset(p, 0, n * sizeof(T)); return p; }

int main(int argc, char** argv) {
    int C = argc > 1 ? atoi(argv[1]) : 4096;
    int m = argc > 2 ? atoi(argv[2]) : 1;
    int iters = argc > 3 ? atoi(argv[3]) : 10;
    LaunchCtx ctx{};
    Dims& d = ctx.d; d.C = C; d.m = m; d.N = 16384 * m; d.n_fm_in = 16384; d.n_fm_out = 8192; d.n_rds = 1024; d.n_audio = 2048; d.n_est = 205;
    d.tail_base = front_tail_len(m);
    ctx.bytes_cap = 16 * (d.n_rds / 256 + 1);
    Buffers& b = ctx.b;
    size_t nin = (size_t)C * d.N;
    float2* in = dalloc<float2>(nin);
    {
        std::vector<float2> h(nin);
        for (size_t i = 0; i < nin; i++) { float ph = 0.37f * (float)(i % 100003) + 0.01f * (float)((i * 7) % 1000); h[i].x = 100.f * cosf(ph) + (float)(i % 7); h[i].y = 100.f * sinf(ph) - (float)(i % 5); }
        hipMemcpy(in, h.data(), nin * 8, hipMemcpyHostToDevice);
    }
    for (int p = 0; p < 2; p++) {
        b.base_tail[p] = dalloc<float2>((size_t)C * d.tail_base); b.pre_tail[p] = dalloc<float2>((size_t)C * 64); b.fm_in[p] = dalloc<float2>((size_t)C * d.n_fm_in); b.iq_tail[p] = dalloc<float2>((size_t)C * 128); b.dt_tail[p] = dalloc<float>((size_t)C * 128);
        b.fo_tail[p] = dalloc<float>((size_t)C * 64); b.fm_out_iq[p] = dalloc<float2>((size_t)C * d.n_fm_out); b.fm_out[p] = dalloc<float>((size_t)C * d.n_fm_out);
        b.pll_dt[p] = dalloc<float>((size_t)C * d.n_fm_out); b.audio[p] = dalloc<float>((size_t)C * d.n_audio * 2); b.rds_sym[p] = dalloc<float>((size_t)C * d.n_rds);
        b.rds_raw_sym[p] = dalloc<float2>(4); b.rds_count[p] = dalloc<int>(C); b.lpr[p] = dalloc<float>(4); b.lmr[p] = dalloc<float>(4);
        b.rds_bytes[p] = dalloc<uint8_t>((size_t)C * ctx.bytes_cap); b.rds_bytes_count[p] = dalloc<int>(C);
    }
#if 1
    for (int p = 0; p < 2; p++) b.pilot[p] = dalloc<float2>((size_t)C * d.n_fm_out);
#endif
    for (int p = 0; p < kSlots; p++) b.rds[p] = dalloc<float2>((size_t)C * d.n_rds);
    b.lmr_est = dalloc<float>((size_t)C * d.n_est);
    b.b_lpr = dalloc<float>((size_t)C * 128); b.b_lmr = dalloc<float>((size_t)C * 128); b.deemph = dalloc<float>((size_t)C * 4); b.mix = dalloc<float>((size_t)C * 2);
    b.state = dalloc<float>((size_t)S_NUM_FIELDS * C); b.spec_stats = dalloc<unsigned long long>(8);
    {
        std::vector<float> t((size_t)C * 128); for (size_t i = 0; i < t.size(); i++) t[i] = 0.01f * (float)((i % 128) - 60) / 64.f;
        hipMemcpy(b.b_lpr, t.data(), t.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b.b_lmr, t.data(), t.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> mx((size_t)C * 2); for (int c = 0; c < C; c++) { mx[2 * c] = 2.f; mx[2 * c + 1] = 1.f; }
        hipMemcpy(b.mix, mx.data(), mx.size() * 4, hipMemcpyHostToDevice);
    }
    for (int i = 0; i < 64; i++) { ctx.front.b_fm_in[i] = 0.015f * (float)(1 + (i % 9)); ctx.front.b_fm_out[i] = 0.02f * (float)(1 + (i % 5)); }
    for (int i = 0; i < 32; i++) ctx.front.b_hilbert_odd[i] = 0.6366f / (float)(2 * i - 31);
    for (int i = 0; i < 128; i++) ctx.rds_taps.b[i] = 0.01f;
    ctx.front.fm_gain = 0.2716f;
    ctx.loops = LoopCoeffs{5.4e-5f, -0.9998f, 1.19f, 0.0024f, 0.0024f, 0.995f, 0.27f, 0.27f, 0.46f, 0.0019f, 0.0019f, 0.996f};
    prepare_kernels();
    launch_reset_state(ctx, nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; int id; } stages[] = {{"k_front", 0}, {"k_pilot_power", 1}, {"k_pilot_pll", 2}, {"k_extract", 3}, {"k_rds_sync", 4}};
    auto run_on = [&](int id, int slot, hipStream_t st) {
        switch (id) {
            case 0: launch_stage_front(ctx, SlotRef{slot, slot}, in, false, st); break;
            case 1: launch_stage_power(ctx, SlotRef{slot, slot}, st); break;
            case 2: launch_stage_pll(ctx, SlotRef{slot, slot}, st); break;
            case 3: launch_stage_extract(ctx, SlotRef{slot, slot}, st); break;
            default: launch_stage_rds(ctx, SlotRef{slot, slot}, st); break;
        }
    };
    auto run = [&](int id, int slot) { run_on(id, slot, nullptr); };
    // one full pass so every stage sees realistic data
    for (int w = 0; w < 2; w++) for (auto& s : stages) run(s.id, w & 1);
    hipDeviceSynchronize();
    for (auto& s : stages) {
        hipEventRecord(e0, nullptr);
        for (int it = 0; it < iters; it++) run(s.id, it & 1);
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-14s C=%d m=%d: %.3f ms\n", s.name, C, m, ms / iters);
        if (s.id == 2) { unsigned long long hs[8]; hipMemcpy(hs, b.spec_stats, 64, hipMemcpyDeviceToHost); printf("   alone: cycle counter %llu, realtime ticks %llu -> %.1f MHz if 100 MHz ticks\n", hs[6], hs[7], (double)hs[6] / (double)hs[7] * 100.0); }
    }
    // how much does each stage slow the PLL when it runs concurrently (second stream, back-to-back launches)?
    if (argc > 4) {
        hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipDeviceSynchronize();
        for (auto& co : stages) {
            if (co.id == 2) continue;
            hipEventRecord(e0, s1);
            run_on(2, 0, s1);
            hipEventRecord(e1, s1);
            for (int it = 0; it < 12; it++) run_on(co.id, 1, s2);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipDeviceSynchronize();
            unsigned long long hs[8]; hipMemcpy(hs, b.spec_stats, 64, hipMemcpyDeviceToHost);
            printf("k_pilot_pll with %-14s running beside it: %.3f ms  (cycle counter %llu, realtime ticks %llu -> %.1f MHz if 100 MHz ticks)\n", co.name, ms, hs[6], hs[7], (double)hs[6] / (double)hs[7] * 100.0);
        }
    }
    return 0;
}
