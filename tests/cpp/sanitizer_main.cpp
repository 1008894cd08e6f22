// CPU sanitizer driver ("sanitizers on the CPU build"; built by `make -C oracle asan` with -fsanitize=address,undefined).
// TEST INFRASTRUCTURE: runs the oracle (oracle/fm_oracle.c) and the library's host-side filter designer (fm-radio_amd/csrc/fmd_design.cpp)
// under AddressSanitizer / UBSan on a capture file, and writes what they produce for tests/test_sanitizers_cpu.py to compare with the golden
// fixtures.  The reference's own known trouble spots on this path are a static lambda capture in its designer
// (src/dsp/filter_designer.cpp:235,288,345) and the re-blocking buffer's span arithmetic (src/utility/reconstruction_buffer.h:16-26).
//
//   sanitizer_main chain <capture.u8|capture.cf32> <u8|cf32> <block_size> <fs_baseband> <out_prefix>
//       -> <out_prefix>.audio.f32, .rds_sym.f32, .rds_count.i32, .lmr_phase.f32 (per block, concatenated), ragged re-blocking included:
//          the capture is fed in pieces of 1000 + 37 k samples through a reconstruction buffer as src/app.cpp:39-50 does
//   sanitizer_main design <fs_baseband> <out.coeffs>      -> the fmd_coeffs struct as designed by fmd_design.cpp (default controls)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
#include "fm_oracle.h"
}
#include "fmd_design.h"

static std::vector<unsigned char> slurp(const char* p) {
    FILE* f = fopen(p, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", p); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> v((size_t)n);
    if (n && fread(v.data(), 1, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}
static void dump(const std::string& path, const void* p, size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || (bytes && fwrite(p, 1, bytes, f) != bytes)) exit(3);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc >= 4 && std::string(argv[1]) == "design") {
        fmd_controls c{};
        c.audio_out = FMD_AUDIO_STEREO; c.audio_stereo_mix_factor = 1.0f; c.use_deemphasis = 0; c.deemphasis_tus = 1; c.lpr_cutoff_hz = 15000; c.lmr_cutoff_hz = 15000;
        fmd_coeffs k;
        std::memset(&k, 0, sizeof(k));
        fmd::design_all(&k, atoi(argv[2]), &c);
        c.use_deemphasis = 1; c.deemphasis_tus = 50; c.lpr_cutoff_hz = 12000;       // the controls-dependent subset, redesigned
        fmd_coeffs k2 = k;
        fmd::design_controls(&k2, &c);
        dump(argv[3], &k, sizeof(k));
        dump(std::string(argv[3]) + ".ctl", &k2, sizeof(k2));
        return 0;
    }
    if (argc < 7 || std::string(argv[1]) != "chain") { fprintf(stderr, "usage: see the header comment\n"); return 1; }
    const std::vector<unsigned char> cap = slurp(argv[2]);
    const bool u8 = std::string(argv[3]) == "u8";
    const int bs = atoi(argv[4]), fs = atoi(argv[5]);
    const std::string out = argv[6];
    const size_t sample_bytes = u8 ? 2 : 8, n_samples = cap.size() / sample_bytes;
    fmo_demod* d = fmo_create(bs, fs);
    if (!d) return 4;
    std::vector<unsigned char> block((size_t)bs * sample_bytes);
    size_t fill = 0, pos = 0, piece = 1000;
    std::vector<float> audio, syms, phase;
    std::vector<int> counts;
    while (pos < n_samples) {                                   // ragged pieces -> whole blocks (ReconstructionBuffer semantics)
        size_t n = std::min(piece, n_samples - pos);
        piece += 37;
        while (n) {
            const size_t take = std::min(n, (size_t)bs - fill);
            std::memcpy(block.data() + fill * sample_bytes, cap.data() + pos * sample_bytes, take * sample_bytes);
            fill += take; pos += take; n -= take;
            if (fill == (size_t)bs) {
                const int rc = u8 ? fmo_process_u8(d, block.data(), bs) : fmo_process_cf32(d, reinterpret_cast<const float*>(block.data()), bs);
                if (rc != 0) return 5;
                int m = 0;
                const float* a = fmo_get(d, "audio", &m); audio.insert(audio.end(), a, a + m);
                const float* s = fmo_get(d, "rds_sym", &m); syms.insert(syms.end(), s, s + m);
                counts.push_back(fmo_rds_symbol_count(d));
                const float* p = fmo_get(d, "lmr_phase", &m); phase.insert(phase.end(), p, p + m);
                fill = 0;
            }
        }
    }
    if (fmo_process_u8(d, block.data(), bs - 1) != -1) return 6;    // a wrong-sized block is refused, nothing emitted (broadcast_fm_demod.cpp:311-313)
    fmo_destroy(d);
    dump(out + ".audio.f32", audio.data(), audio.size() * 4);
    dump(out + ".rds_sym.f32", syms.data(), syms.size() * 4);
    dump(out + ".rds_count.i32", counts.data(), counts.size() * 4);
    dump(out + ".lmr_phase.f32", phase.data(), phase.size() * 4);
    return 0;
}
