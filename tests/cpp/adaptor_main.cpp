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
    fmd_host::App_GPU app(block_size);
    auto& demod = app.GetFMDemod();
    app.OnAudioBlock().Attach([&](const fmd_host::Frame* x, size_t n, int Fs) {
        (void)Fs;
        fwrite(x, sizeof(fmd_host::Frame), n, fa);
        auto l = demod.GetLPRAudioOutput();
        fwrite(l.data(), sizeof(float), l.size(), fl);
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
    fclose(fa); fclose(fs); fclose(fb); fclose(fl);
    printf("%zu blocks of %d\n", n_blocks, block_size);
    return 0;
}
