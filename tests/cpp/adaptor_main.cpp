// Test driver written the way the reference's fm_demod_benchmark / dump harness uses App (reference
// src/fm_demod_benchmark.cpp:90-101): feed a u8 capture in odd-sized pieces, collect what the observers deliver.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "broadcast_fm_demod_gpu.hpp"

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: adaptor_main <capture.u8> <outdir> <block_size>\n"); return 1; }
    const int block_size = atoi(argv[3]);
    FILE* fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    fseek(fp, 0, SEEK_END); long bytes = ftell(fp); fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> data((size_t)bytes);
    if (fread(data.data(), 1, data.size(), fp) != data.size()) return 2;
    fclose(fp);
    std::string out = argv[2];
    FILE* fa = fopen((out + "/audio.f32").c_str(), "wb");
    FILE* fs = fopen((out + "/rds_sym.f32").c_str(), "wb");
    FILE* fb = fopen((out + "/rds_bytes.u8").c_str(), "wb");
    FILE* fl = fopen((out + "/lpr.f32").c_str(), "wb");
    // the loops' per-sample getters (reference broadcast_fm_demod.h:245-248, bpsk_synchroniser.h:78-85), as a GUI would read them per block
    FILE* fp1 = fopen((out + "/pll.cf32").c_str(), "wb");
    FILE* fp2 = fopen((out + "/pll_pi_err.f32").c_str(), "wb");
    FILE* fp3 = fopen((out + "/pilot.cf32").c_str(), "wb");
    FILE* fp4 = fopen((out + "/bpsk_ted_pi.f32").c_str(), "wb");
    FILE* fp5 = fopen((out + "/bpsk_pll_sym.cf32").c_str(), "wb");
    fmd_host::App_GPU app(block_size);
    auto& demod = app.GetFMDemod();
    app.OnAudioBlock().Attach([&](const fmd_host::Frame* x, size_t n, int Fs) {
        (void)Fs;
        fwrite(x, sizeof(fmd_host::Frame), n, fa);
        auto l = demod.GetLPRAudioOutput();
        fwrite(l.data(), sizeof(float), l.size(), fl);
        auto p1 = demod.GetPLLOutput();                        fwrite(p1.data(), sizeof(std::complex<float>), p1.size(), fp1);
        auto p2 = demod.Get_PLL_LPF_Phase_Error_Output();      fwrite(p2.data(), sizeof(float), p2.size(), fp2);
        auto p3 = demod.GetPilotOutput();                      fwrite(p3.data(), sizeof(std::complex<float>), p3.size(), fp3);
        auto sync = demod.GetBPSKSync();
        auto p4 = sync.GetTEDPIPhaseError();                   fwrite(p4.data(), sizeof(float), p4.size(), fp4);
        auto p5 = sync.GetPLLSymbols();                        fwrite(p5.data(), sizeof(std::complex<float>), p5.size(), fp5);
    });
    demod.OnRDSOut().Attach([&](const float* x, size_t n) { fwrite(x, sizeof(float), n, fs); });
    app.On_RDS_Bytes().Attach([&](const uint8_t* x, size_t n) { fwrite(x, 1, n, fb); });
    const size_t n_samples = data.size() / 2, n_blocks = n_samples / block_size, total = n_blocks * (size_t)block_size;
    size_t pos = 0;
    const size_t piece = 16384 + 123;
    while (pos < total) {
        size_t n = (total - pos < piece) ? (total - pos) : piece;
        app.Process(data.data() + 2 * pos, n);
        pos += n;
    }
    fclose(fa); fclose(fs); fclose(fb); fclose(fl); fclose(fp1); fclose(fp2); fclose(fp3); fclose(fp4); fclose(fp5);
    printf("%zu blocks of %d\n", n_blocks, block_size);
    return 0;
}
