// Feeds recorded demodulator outputs (float audio blocks, RDS bytes) through the scraper-compatible writers.
//   scraper_writer_main <audio.f32> <rds_bytes.u8> <n_frames_per_block> <out.wav> <out_rds.bin>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fm_scraper_writer.hpp"

static std::vector<unsigned char> slurp(const char* p) {
    FILE* f = fopen(p, "rb"); if (!f) exit(2);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> v((size_t)n); if (n && fread(v.data(), 1, v.size(), f) != v.size()) exit(2);
    fclose(f); return v;
}
int main(int argc, char** argv) {
    if (argc < 6) return 1;
    auto a = slurp(argv[1]); auto r = slurp(argv[2]);
    const size_t per = (size_t)atoi(argv[3]);
    const float* audio = reinterpret_cast<const float*>(a.data());
    const size_t frames = a.size() / 8;
    {
        fmd_host::Audio_WAV_Writer w(argv[4], 32000);
        for (size_t i = 0; i + per <= frames; i += per) w.on_audio_data(audio + 2 * i, per);
        fmd_host::RDS_Bytes_Writer rb(argv[5]);
        for (size_t i = 0; i + 16 <= r.size(); i += 16) rb.on_rds_bytes(r.data() + i, 16);
    }
    return 0;
}
