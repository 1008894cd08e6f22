// TEST INFRASTRUCTURE — dump harness around the *unmodified* reference sources.
//
// This file is ours; it is compiled together with the reference's own translation
// units where they lie under /root/reference/src (see oracle/Makefile) and only
// ever produces files under oracle/_ref/.  It drives the reference through its
// public API and writes raw little-endian dumps of the buffers the reference
// exposes through its getters, so the C restatement in oracle/fm_oracle.c (and
// through it the HIP path) can be pinned against the real thing.
//
//   fm_ref_dump chain     <capture.u8>   <outdir> <block_size> [deemph=us audio=lpr|lmr|stereo mix=f lpr=hz lmr=hz]   App::Process path (u8 ingest)
//   fm_ref_dump cf32chain <capture.cf32> <outdir> <block_size>   Broadcast_FM_Demod::Process path
//   fm_ref_dump prims     <indir>        <outdir>                per-primitive vectors
//
// Reference API used: App (src/app.h:19-47), Broadcast_FM_Demod getters
// (src/fm_demod/broadcast_fm_demod.h:229-298), BPSK_Synchroniser getters
// (src/fm_demod/bpsk_synchroniser.h:76-84), DifferentialManchesterDecoder
// (src/rds_decoder/differential_manchester_decoder.h), filter designer
// (src/dsp/filter_designer.h) and the header-only DSP templates in src/dsp.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <complex>
#include <string>
#include <vector>
#include <memory>

#include "app.h"
#include "fm_demod/broadcast_fm_demod.h"
#include "fm_demod/bpsk_synchroniser.h"
#include "fm_demod/fm_demod.h"
#include "rds_decoder/differential_manchester_decoder.h"
#include "dsp/filter_designer.h"
#include "dsp/polyphase_filter.h"
#include "dsp/hilbert_fir_filter.h"
#include "dsp/iir_filter.h"
#include "dsp/agc.h"
#include "dsp/simd/apply_harmonic_pll.h"
#include "dsp/simd/chebyshev_sine.h"

typedef std::complex<float> cf32;

struct Sink {
    std::string dir;
    FILE* open(const char* name) {
        std::string p = dir + "/" + name;
        FILE* fp = fopen(p.c_str(), "wb");
        if (!fp) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
        return fp;
    }
};

template <typename T>
static void put(FILE* fp, const T* x, size_t n) { fwrite((const void*)x, sizeof(T), n, fp); }

template <typename T>
static std::vector<T> slurp(const std::string& path) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T));
    if (fread((void*)v.data(), sizeof(T), v.size(), fp) != v.size()) { fprintf(stderr, "short read %s\n", path.c_str()); exit(2); }
    fclose(fp);
    return v;
}

struct ChainDump {
    FILE *fm_out_iq, *pilot, *pll, *pll_raw, *pll_pi, *lpr, *lmr, *rds, *rds_raw_sym, *rds_sym, *rds_count;
    FILE *audio, *lmr_phase, *rds_bytes;
    FILE *bpsk_pll_sym, *bpsk_zcd, *bpsk_trig, *bpsk_ted_raw, *bpsk_ted_pi, *bpsk_pll_raw, *bpsk_pll_pi, *bpsk_intdump;
    explicit ChainDump(Sink& s) {
        fm_out_iq = s.open("fm_out_iq.cf32"); pilot = s.open("pilot.cf32"); pll = s.open("pll.cf32");
        pll_raw = s.open("pll_raw_err.f32"); pll_pi = s.open("pll_pi_err.f32");
        lpr = s.open("lpr.f32"); lmr = s.open("lmr.f32"); rds = s.open("rds.cf32");
        rds_raw_sym = s.open("rds_raw_sym.cf32"); rds_sym = s.open("rds_sym.f32"); rds_count = s.open("rds_count.i32");
        audio = s.open("audio.f32"); lmr_phase = s.open("lmr_phase.f32"); rds_bytes = s.open("rds_bytes.u8");
        bpsk_pll_sym = s.open("bpsk_pll_sym.cf32"); bpsk_zcd = s.open("bpsk_zcd.u8"); bpsk_trig = s.open("bpsk_trig.u8");
        bpsk_ted_raw = s.open("bpsk_ted_raw.f32"); bpsk_ted_pi = s.open("bpsk_ted_pi.f32");
        bpsk_pll_raw = s.open("bpsk_pll_raw.f32"); bpsk_pll_pi = s.open("bpsk_pll_pi.f32"); bpsk_intdump = s.open("bpsk_intdump.cf32");
    }
    void on_block(Broadcast_FM_Demod& d, tcb::span<const Frame<float>> a) {
        auto v0 = d.GetFMOutIQ();                      put(fm_out_iq, v0.data(), v0.size());
        auto v1 = d.GetPilotOutput();                  put(pilot, v1.data(), v1.size());
        auto v2 = d.GetPLLOutput();                    put(pll, v2.data(), v2.size());
        auto v3 = d.Get_PLL_Raw_Phase_Error_Output();  put(pll_raw, v3.data(), v3.size());
        auto v4 = d.Get_PLL_LPF_Phase_Error_Output();  put(pll_pi, v4.data(), v4.size());
        auto v5 = d.GetLPRAudioOutput();               put(lpr, v5.data(), v5.size());
        auto v6 = d.GetLMRAudioOutput();               put(lmr, v6.data(), v6.size());
        auto v7 = d.GetRDSOutput();                    put(rds, v7.data(), v7.size());
        auto v8 = d.GetRDSRawSymbols();                put(rds_raw_sym, v8.data(), v8.size());
        auto v9 = d.GetRDSPredSymbols();               put(rds_sym, v9.data(), v9.size());
        int32_t n = (int32_t)v9.size();                put(rds_count, &n, 1);
        put(audio, reinterpret_cast<const float*>(a.data()), a.size() * 2);
        float ph = d.GetAudioLMRPhaseError();          put(lmr_phase, &ph, 1);
        auto& b = d.GetBPSKSync();
        auto b0 = b.GetPLLSymbols();      put(bpsk_pll_sym, b0.data(), b0.size());
        auto b1 = b.GetZeroCrossings();   put(bpsk_zcd, reinterpret_cast<const uint8_t*>(b1.data()), b1.size());
        auto b2 = b.GetIntDumpTriggers(); put(bpsk_trig, reinterpret_cast<const uint8_t*>(b2.data()), b2.size());
        auto b3 = b.GetTEDRawPhaseError(); put(bpsk_ted_raw, b3.data(), b3.size());
        auto b4 = b.GetTEDPIPhaseError();  put(bpsk_ted_pi, b4.data(), b4.size());
        auto b5 = b.GetPLLRawPhaseError(); put(bpsk_pll_raw, b5.data(), b5.size());
        auto b6 = b.GetPLLPIPhaseError();  put(bpsk_pll_pi, b6.data(), b6.size());
        auto b7 = b.GetIntDumpFilter();    put(bpsk_intdump, b7.data(), b7.size());
    }
    void close_all() {
        FILE* all[] = { fm_out_iq, pilot, pll, pll_raw, pll_pi, lpr, lmr, rds, rds_raw_sym, rds_sym, rds_count, audio, lmr_phase, rds_bytes,
                        bpsk_pll_sym, bpsk_zcd, bpsk_trig, bpsk_ted_raw, bpsk_ted_pi, bpsk_pll_raw, bpsk_pll_pi, bpsk_intdump };
        for (FILE* f : all) fclose(f);
    }
};

// optional key=value arguments -> Broadcast_FM_Demod_Controls (reference broadcast_fm_demod.h:64-89)
static void apply_controls(Broadcast_FM_Demod& d, int argc, char** argv) {
    auto& c = d.GetControls();
    for (int i = 0; i < argc; i++) {
        const char* a = argv[i];
        if (!strncmp(a, "deemph=", 7)) { c.is_use_deemphasis_filter = true; c.filt_deemphasis_cutoff.SetValue(atoi(a + 7)); }
        else if (!strncmp(a, "audio=", 6)) {
            if (!strcmp(a + 6, "lpr")) c.audio_out = Broadcast_FM_Demod_Controls::LPR;
            else if (!strcmp(a + 6, "lmr")) c.audio_out = Broadcast_FM_Demod_Controls::LMR;
            else c.audio_out = Broadcast_FM_Demod_Controls::STEREO;
        }
        else if (!strncmp(a, "mix=", 4)) c.audio_stereo_mix_factor = (float)atof(a + 4);
        else if (!strncmp(a, "lpr=", 4)) c.filt_audio_lpr_cutoff.SetValue(atoi(a + 4));
        else if (!strncmp(a, "lmr=", 4)) c.filt_audio_lmr_cutoff.SetValue(atoi(a + 4));
        else { fprintf(stderr, "unknown control %s\n", a); exit(2); }
    }
}

static int run_chain(const char* in, const char* outdir, int block_size, int argc, char** argv) {
    Sink sink{outdir};
    ChainDump dump(sink);
    auto data = slurp<std::complex<uint8_t>>(in);
    App app(block_size);
    auto& demod = app.GetFMDemod();
    apply_controls(demod, argc, argv);
    app.OnAudioBlock().Attach([&](tcb::span<const Frame<float>> x, const int Fs) { (void)Fs; dump.on_block(demod, x); });
    app.On_RDS_Bytes().Attach([&](tcb::span<const uint8_t> x) { put(dump.rds_bytes, x.data(), x.size()); });
    const size_t n_blocks = data.size() / (size_t)block_size;
    // feed in odd-sized pieces to exercise the reference's re-blocking (src/app.cpp:39-50)
    const size_t total = n_blocks * (size_t)block_size;
    size_t pos = 0, piece = 16384;
    while (pos < total) {
        size_t n = (total - pos < piece) ? (total - pos) : piece;
        app.Process(tcb::span<const std::complex<uint8_t>>(data.data() + pos, n));
        pos += n;
    }
    dump.close_all();
    fprintf(stderr, "[ref_dump] chain: %zu blocks of %d\n", n_blocks, block_size);
    return 0;
}

static int run_cf32chain(const char* in, const char* outdir, int block_size, int argc, char** argv) {
    Sink sink{outdir};
    ChainDump dump(sink);
    auto data = slurp<cf32>(in);
    Broadcast_FM_Demod demod(block_size);
    apply_controls(demod, argc, argv);
    uint8_t bytes_buf[16];
    DifferentialManchesterDecoder manchester{tcb::span<uint8_t>(bytes_buf, 16)};
    demod.OnAudioOut().Attach([&](tcb::span<const Frame<float>> x, const int Fs) { (void)Fs; dump.on_block(demod, x); });
    demod.OnRDSOut().Attach([&](tcb::span<const float> x) { manchester.Process(x); });
    manchester.OnBytes().Attach([&](tcb::span<const uint8_t> x) { put(dump.rds_bytes, x.data(), x.size()); });
    const size_t n_blocks = data.size() / (size_t)block_size;
    for (size_t b = 0; b < n_blocks; b++) {
        demod.Process(tcb::span<const cf32>(data.data() + b * (size_t)block_size, (size_t)block_size));
    }
    dump.close_all();
    fprintf(stderr, "[ref_dump] cf32chain: %zu blocks of %d\n", n_blocks, block_size);
    return 0;
}

// stream `x` through a filter in three unequal pieces so the history carry is exercised
template <typename F, typename Tin, typename Tout>
static std::vector<Tout> stream3(F& filt, const std::vector<Tin>& x, int M, size_t n_out_total) {
    std::vector<Tout> y(n_out_total);
    size_t cuts[4] = { 0, n_out_total / 4, n_out_total / 4 + n_out_total / 8, n_out_total };
    for (int p = 0; p < 3; p++) {
        size_t o0 = cuts[p], o1 = cuts[p + 1];
        filt.process(x.data() + o0 * (size_t)M, y.data() + o0, (int)(o1 - o0));
    }
    return y;
}

static int run_prims(const char* indir, const char* outdir) {
    Sink s{outdir};
    std::string in = indir;
    auto xc = slurp<cf32>(in + "/x_c.cf32");    // complex noise-like input
    auto xr = slurp<float>(in + "/x_r.f32");    // real input
    auto grid = slurp<float>(in + "/grid.f32"); // points in [-0.5, 0.5]
    auto dt = slurp<float>(in + "/dt.f32");     // pll phase ramp

    // 1. filter designs used by the chain (src/fm_demod/broadcast_fm_demod.cpp:133-274,330-389; bpsk_synchroniser.cpp:26-48)
    {
        FILE* fp = s.open("taps.f32");
        float b64[64], b128[128], b65[65], b2[2], a2[2], b3[3], a3[3];
        create_fir_lpf(b64, 64, 0.25f * 0.95f);  put(fp, b64, 64);      // stage 1, fs 1.024M -> 256k
        create_fir_lpf(b64, 64, 0.5f * 0.95f);   put(fp, b64, 64);      // stage 2, 256k -> 128k
        create_fir_lpf(b64, 64, 0.125f * 0.95f); put(fp, b64, 64);      // stage 1 for fs 2.048M (M=8)
        create_fir_lpf(b128, 128, 15000.0f / 64000.0f); put(fp, b128, 128); // lpr / lmr default
        create_fir_lpf(b128, 128, 2000.0f / 64000.0f);  put(fp, b128, 128); // rds
        create_fir_hilbert(b65, 65);             put(fp, b65, 65);
        create_iir_peak_1_filter(b3, a3, 19000.0f / 64000.0f, 0.9999f); put(fp, b3, 3); put(fp, a3, 3);
        create_iir_single_pole_lpf(b2, a2, 100.0f / 64000.0f);  put(fp, b2, 2); put(fp, a2, 2);  // pilot pll loop
        create_iir_single_pole_lpf(b2, a2, 1500.0f / 8000.0f);  put(fp, b2, 2); put(fp, a2, 2);  // bpsk ted loop
        create_iir_single_pole_lpf(b2, a2, 10.0f / 8000.0f);    put(fp, b2, 2); put(fp, a2, 2);  // bpsk pll loop
        // de-emphasis designs for 1, 50, 75 us (broadcast_fm_demod.cpp:337-352)
        const int tus[3] = { 1, 50, 75 };
        for (int i = 0; i < 3; i++) {
            const float Tc = (float)tus[i] * 1e-6f;
            const float Fc = 1.0f / (2.0f * (float)M_PI * Tc);
            float k = Fc / (128000.0f / 2.0f);
            k = (k > 0.01f) ? k : 0.01f; k = (k > 0.99f) ? 0.99f : k;
            create_iir_single_pole_lpf(b2, a2, k); put(fp, b2, 2); put(fp, a2, 2);
        }
        // audio cut-off clamp extremes (broadcast_fm_demod.cpp:332-334)
        create_fir_lpf(b128, 128, 0.01f); put(fp, b128, 128);
        create_fir_lpf(b128, 128, 0.99f); put(fp, b128, 128);
        fclose(fp);
    }
    // 2. decimating FIRs with history carry (src/dsp/polyphase_filter.h:41-64)
    {
        PolyphaseDownsampler<cf32> f1(4, 16); create_fir_lpf(f1.get_b(), f1.get_K(), 0.2375f);
        auto y1 = stream3<PolyphaseDownsampler<cf32>, cf32, cf32>(f1, xc, 4, xc.size() / 4);
        FILE* fp = s.open("poly_c_m4_n64.cf32"); put(fp, y1.data(), y1.size()); fclose(fp);

        PolyphaseDownsampler<cf32> f8(8, 8); create_fir_lpf(f8.get_b(), f8.get_K(), 0.11875f);
        auto y8 = stream3<PolyphaseDownsampler<cf32>, cf32, cf32>(f8, xc, 8, xc.size() / 8);
        fp = s.open("poly_c_m8_n64.cf32"); put(fp, y8.data(), y8.size()); fclose(fp);

        PolyphaseDownsampler<float> f2(2, 32); create_fir_lpf(f2.get_b(), f2.get_K(), 0.475f);
        auto y2 = stream3<PolyphaseDownsampler<float>, float, float>(f2, xr, 2, xr.size() / 2);
        fp = s.open("poly_r_m2_n64.f32"); put(fp, y2.data(), y2.size()); fclose(fp);

        PolyphaseDownsampler<cf32> f3(4, 32); create_fir_lpf(f3.get_b(), f3.get_K(), 15000.0f / 64000.0f);
        auto y3 = stream3<PolyphaseDownsampler<cf32>, cf32, cf32>(f3, xc, 4, xc.size() / 4);
        fp = s.open("poly_c_m4_n128.cf32"); put(fp, y3.data(), y3.size()); fclose(fp);

        PolyphaseDownsampler<cf32> f4(8, 16); create_fir_lpf(f4.get_b(), f4.get_K(), 2000.0f / 64000.0f);
        auto y4 = stream3<PolyphaseDownsampler<cf32>, cf32, cf32>(f4, xc, 8, xc.size() / 8);
        fp = s.open("poly_c_m8_n128.cf32"); put(fp, y4.data(), y4.size()); fclose(fp);
    }
    // 3. Hilbert FIR (src/dsp/hilbert_fir_filter.h:26-46)
    {
        Hilbert_FIR_Filter<float> h(65);
        auto y = stream3<Hilbert_FIR_Filter<float>, float, cf32>(h, xr, 1, xr.size());
        FILE* fp = s.open("hilbert.cf32"); put(fp, y.data(), y.size()); fclose(fp);
    }
    // 4. IIR filters (src/dsp/iir_filter.h:40-69)
    {
        IIR_Filter<cf32> pk(3); create_iir_peak_1_filter(pk.get_b(), pk.get_a(), 19000.0f / 64000.0f, 0.9999f);
        auto y = stream3<IIR_Filter<cf32>, cf32, cf32>(pk, xc, 1, xc.size());
        FILE* fp = s.open("iir_peak.cf32"); put(fp, y.data(), y.size()); fclose(fp);
        IIR_Filter<float> lp(2); create_iir_single_pole_lpf(lp.get_b(), lp.get_a(), 0.0497f);
        auto y2 = stream3<IIR_Filter<float>, float, float>(lp, xr, 1, xr.size());
        fp = s.open("iir_lpf.f32"); put(fp, y2.data(), y2.size()); fclose(fp);
    }
    // 5. AGC, three consecutive blocks (src/dsp/agc.h:12-30)
    {
        AGC_Filter<cf32> agc; agc.target_power = 0.5f;
        std::vector<cf32> y(xc.size());
        size_t nb = xc.size() / 3;
        FILE* fg = s.open("agc_gain.f32");
        for (int b = 0; b < 3; b++) { agc.process(xc.data() + b * nb, y.data() + b * nb, (int)nb); put(fg, &agc.current_gain, 1); }
        fclose(fg);
        FILE* fp = s.open("agc.cf32"); put(fp, y.data(), nb * 3); fclose(fp);
    }
    // 6. FM discriminator, two blocks (src/fm_demod/fm_demod.cpp:30-45)
    {
        FM_Demod d;
        std::vector<float> y(xc.size());
        size_t h = xc.size() / 2;
        d.Process(tcb::span<const cf32>(xc.data(), h), tcb::span<float>(y.data(), h), 75e3f, 256000.0f);
        d.Process(tcb::span<const cf32>(xc.data() + h, xc.size() - h), tcb::span<float>(y.data() + h, xc.size() - h), 75e3f, 256000.0f);
        FILE* fp = s.open("fm_discriminator.f32"); put(fp, y.data(), y.size()); fclose(fp);
    }
    // 7. chebyshev sine (src/dsp/simd/chebyshev_sine.h:22-41)
    {
        std::vector<float> y(grid.size());
        for (size_t i = 0; i < grid.size(); i++) y[i] = chebyshev_sine(grid[i]);
        FILE* fp = s.open("chebyshev.f32"); put(fp, y.data(), y.size()); fclose(fp);
    }
    // 8. harmonic mixer (src/dsp/simd/apply_harmonic_pll.cpp:144-159); odd length exercises the scalar tail
    {
        size_t n = dt.size() - 1;
        std::vector<cf32> y2(n), y3(n);
        apply_harmonic_pll_auto(dt.data(), xc.data(), y2.data(), (int)n, 2.0f, 0.3217f);
        apply_harmonic_pll_auto(dt.data(), xc.data(), y3.data(), (int)n, 3.0f, 0.0f);
        FILE* fp = s.open("harmonic2.cf32"); put(fp, y2.data(), n); fclose(fp);
        fp = s.open("harmonic3.cf32"); put(fp, y3.data(), n); fclose(fp);
    }
    fprintf(stderr, "[ref_dump] prims done\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 5 && !strcmp(argv[1], "chain")) return run_chain(argv[2], argv[3], atoi(argv[4]), argc - 5, argv + 5);
    if (argc >= 5 && !strcmp(argv[1], "cf32chain")) return run_cf32chain(argv[2], argv[3], atoi(argv[4]), argc - 5, argv + 5);
    if (argc >= 4 && !strcmp(argv[1], "prims")) return run_prims(argv[2], argv[3]);
    fprintf(stderr, "usage: fm_ref_dump chain|cf32chain <in> <outdir> <block> | prims <indir> <outdir>\n");
    return 1;
}
