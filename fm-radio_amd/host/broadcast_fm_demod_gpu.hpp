// Header-only C++ host adaptor over the C ABI (include/fmdemod.h) with the reference's demodulator interface, so code
// written against williamyang98/FM-Radio's `Broadcast_FM_Demod` / `App` can switch to the MI355X library:
//
//   reference (src/fm_demod/broadcast_fm_demod.h:229-298)        this adaptor
//   Broadcast_FM_Demod(int block_size)                            Broadcast_FM_Demod_GPU(block_size[, n_channels, fs])
//   void Process(span<const complex<float>>)                      void Process(const std::complex<float>* x, size_t n)
//   OnAudioOut().Attach(fn(span<const Frame<float>>, int Fs))     OnAudioOut().Attach(fn(const Frame*, size_t n, int Fs))
//   OnRDSOut().Attach(fn(span<const float>))                      OnRDSOut().Attach(fn(const float*, size_t n))
//   GetAudioOut() / GetRDSPredSymbols() / GetLPRAudioOutput() ... same names, return pointer+size views
//   GetPilotOutput() / GetPLLOutput() / Get_PLL_*_Phase_Error_Output() (.h:244-248), GetBPSKSync().Get*() (bpsk_synchroniser.h:78-85): same names
//   GetControls()                                                 GetControls() / ApplyControls()
//   App::Process(span<const complex<uint8_t>>) (src/app.cpp:39-50)  App_GPU::Process(const uint8_t* iq, size_t n_samples)
//
// Like the reference, observers run synchronously on the caller's thread inside Process (utility/observable.h:17-21),
// wrong-sized blocks are dropped silently (broadcast_fm_demod.cpp:311-313) and outputs are views valid until the next
// Process.  With n_channels > 1, x is [C][block_size] and observers receive channel 0.. in order (one call per channel).
#pragma once

#include <complex>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "fmdemod.h"

namespace fmd_host {

#ifndef FMD_HOST_FRAME_DEFINED
#define FMD_HOST_FRAME_DEFINED
struct Frame { float channels[2]; };  // reference src/audio/frame.h:6-8
#endif

template <typename... T>
class Observable {  // reference src/utility/observable.h:7-22
    std::vector<std::function<void(T...)>> observers;
public:
    void Attach(const std::function<void(T...)>& o) { observers.push_back(o); }
    void Notify(T... args) { for (auto& o : observers) o(args...); }
};

template <typename T> struct View { const T* ptr; size_t n; const T* data() const { return ptr; } size_t size() const { return n; }
                                     const T& operator[](size_t i) const { return ptr[i]; } };

class Broadcast_FM_Demod_GPU {
    fmd_handle h = nullptr;
    fmd_rates rates{};
    fmd_controls controls{};
    int n_channels, block_size;
    std::vector<float> audio, rds_sym, lpr, lmr, fm_out_iq, rds, rds_raw;
    std::vector<float> pilot, pll, pll_raw_err, pll_pi_err, b_pll_sym, b_intdump, b_zcd, b_trig, b_ted_raw, b_ted_pi, b_pll_raw, b_pll_pi;
    std::vector<int> rds_count;
    std::vector<uint8_t> rds_bytes;
    std::vector<int> rds_bytes_count;
    int bytes_cap;
    Observable<const Frame*, size_t, int> obs_on_audio_block;
    Observable<const float*, size_t> obs_on_rds_symbols;
    Observable<const uint8_t*, size_t> obs_on_rds_bytes;
    void check(int rc, const char* what) { if (rc != FMD_OK) throw std::runtime_error(std::string(what) + ": " + fmd_last_error(h)); }
    void fetch(const char* name, std::vector<float>& v) {
        size_t n = 0;
        fmd_get_stream(h, name, v.data(), 0, &n);
        v.resize(n);
        check(fmd_get_stream(h, name, v.data(), v.size(), &n), name);
    }
    View<float> fview(const char* name, std::vector<float>& v, int c, int n) { fetch(name, v); return {v.data() + (size_t)c * n, (size_t)n}; }
    View<std::complex<float>> cview(const char* name, std::vector<float>& v, int c, int n) {
        fetch(name, v);
        return {reinterpret_cast<const std::complex<float>*>(v.data()) + (size_t)c * n, (size_t)n};
    }
public:
    explicit Broadcast_FM_Demod_GPU(int _block_size, int _n_channels = 1, int fs_baseband = 1024000, bool keep_taps = true)
        : n_channels(_n_channels), block_size(_block_size) {
        fmd_config cfg{_n_channels, _block_size, fs_baseband, -1, keep_taps ? FMD_FLAG_KEEP_TAPS : 0u};
        int rc = fmd_create(&cfg, &h);
        if (rc != FMD_OK) throw std::runtime_error(std::string("fmd_create: ") + fmd_last_error(nullptr));
        fmd_get_rates(h, &rates);
        fmd_default_controls(&controls);
        audio.resize((size_t)n_channels * rates.n_audio * 2);
        rds_sym.resize((size_t)n_channels * rates.n_rds);
        rds_count.assign(n_channels, 0);
        bytes_cap = 16 * (rates.n_rds / 256 + 1);
        rds_bytes.resize((size_t)n_channels * bytes_cap);
        rds_bytes_count.assign(n_channels, 0);
    }
    ~Broadcast_FM_Demod_GPU() { if (h) fmd_destroy(h); }
    Broadcast_FM_Demod_GPU(const Broadcast_FM_Demod_GPU&) = delete;
    Broadcast_FM_Demod_GPU& operator=(const Broadcast_FM_Demod_GPU&) = delete;

    void Process(const std::complex<float>* x, size_t n) { Run(reinterpret_cast<const float*>(x), nullptr, n); }
    void ProcessU8(const uint8_t* iq, size_t n) { Run(nullptr, iq, n); }

    // 4. RDS synchronisation / 5. Audio mixing (reference getters .h:249-256)
    View<Frame> GetAudioOut(int c = 0) const { return {reinterpret_cast<const Frame*>(audio.data()) + (size_t)c * rates.n_audio, (size_t)rates.n_audio}; }
    View<float> GetRDSPredSymbols(int c = 0) const { return {rds_sym.data() + (size_t)c * rates.n_rds, (size_t)rds_count[c]}; }
    View<float> GetLPRAudioOutput(int c = 0) { fetch("lpr", lpr); return {lpr.data() + (size_t)c * rates.n_audio, (size_t)rates.n_audio}; }
    View<float> GetLMRAudioOutput(int c = 0) { fetch("lmr", lmr); return {lmr.data() + (size_t)c * rates.n_audio, (size_t)rates.n_audio}; }
    View<std::complex<float>> GetFMOutIQ(int c = 0) { fetch("fm_out_iq", fm_out_iq);
        return {reinterpret_cast<const std::complex<float>*>(fm_out_iq.data()) + (size_t)c * rates.n_fm_out, (size_t)rates.n_fm_out}; }
    View<std::complex<float>> GetRDSOutput(int c = 0) { fetch("rds", rds);
        return {reinterpret_cast<const std::complex<float>*>(rds.data()) + (size_t)c * rates.n_rds, (size_t)rates.n_rds}; }
    float GetAudioLMRPhaseError(int c = 0) { std::vector<float> v; fetch("lmr_phase", v); return v[c]; }
    // 2. Lock onto pilot (reference .h:244-248) — the loop's per-sample traces; the adaptor runs the exact mode with FMD_FLAG_KEEP_TAPS
    View<std::complex<float>> GetPilotOutput(int c = 0) { return cview("pilot", pilot, c, rates.n_fm_out); }
    View<std::complex<float>> GetPLLOutput(int c = 0) { return cview("pll", pll, c, rates.n_fm_out); }
    View<float> Get_PLL_Raw_Phase_Error_Output(int c = 0) { return fview("pll_raw_err", pll_raw_err, c, rates.n_fm_out); }
    View<float> Get_PLL_LPF_Phase_Error_Output(int c = 0) { return fview("pll_pi_err", pll_pi_err, c, rates.n_fm_out); }
    // BPSK_Synchroniser's views (reference bpsk_synchroniser.h:78-85; GetBPSKSync() of broadcast_fm_demod.h:262): one object per station
    class BPSK_Sync_View {
        Broadcast_FM_Demod_GPU& d; int c;
    public:
        BPSK_Sync_View(Broadcast_FM_Demod_GPU& d_, int c_) : d(d_), c(c_) {}
        View<std::complex<float>> GetPLLSymbols() { return d.cview("bpsk_pll_sym", d.b_pll_sym, c, d.rates.n_rds); }
        View<float> GetZeroCrossings() { return d.fview("bpsk_zcd", d.b_zcd, c, d.rates.n_rds); }            // (0 / 1; the reference's span<const bool>)
        View<float> GetIntDumpTriggers() { return d.fview("bpsk_trig", d.b_trig, c, d.rates.n_rds); }
        View<float> GetTEDRawPhaseError() { return d.fview("bpsk_ted_raw", d.b_ted_raw, c, d.rates.n_rds); }
        View<float> GetTEDPIPhaseError() { return d.fview("bpsk_ted_pi", d.b_ted_pi, c, d.rates.n_rds); }
        View<float> GetPLLRawPhaseError() { return d.fview("bpsk_pll_raw", d.b_pll_raw, c, d.rates.n_rds); }
        View<float> GetPLLPIPhaseError() { return d.fview("bpsk_pll_pi", d.b_pll_pi, c, d.rates.n_rds); }
        View<std::complex<float>> GetIntDumpFilter() { return d.cview("bpsk_intdump", d.b_intdump, c, d.rates.n_rds); }
    };
    BPSK_Sync_View GetBPSKSync(int c = 0) { return BPSK_Sync_View(*this, c); }

    // sample rates (reference .h:284-288)
    int GetBasebandSampleRate() const { return rates.fs_baseband; }
    int GetFMInSampleRate() const { return rates.fs_fm_in; }
    int GetFMOutSampleRate() const { return rates.fs_fm_out; }
    int GetRDSSampleRate() const { return rates.fs_rds; }
    int GetAudioSampleRate() const { return rates.fs_audio; }

    // controls (reference .h:294): edit the returned struct, then ApplyControls(); takes effect at the next block
    fmd_controls& GetControls() { return controls; }
    void ApplyControls(int channel = -1) { check(fmd_set_controls(h, channel, &controls), "fmd_set_controls"); }

    Observable<const Frame*, size_t, int>& OnAudioOut() { return obs_on_audio_block; }
    Observable<const float*, size_t>& OnRDSOut() { return obs_on_rds_symbols; }
    // the Manchester-decoded RDS bytes the reference's App forwards (src/app.cpp:31-34), decoded on the GPU
    Observable<const uint8_t*, size_t>& On_RDS_Bytes() { return obs_on_rds_bytes; }
    fmd_handle Handle() { return h; }

private:
    void Run(const float* cf32, const uint8_t* u8, size_t n) {
        if (n != (size_t)block_size * n_channels) return;  // reference: silently dropped
        int rc = cf32 ? fmd_process_cf32_host(h, cf32, n_channels, block_size) : fmd_process_u8_host(h, u8, n_channels, block_size);
        check(rc, "fmd_process");
        check(fmd_get_audio(h, audio.data()), "fmd_get_audio");
        check(fmd_get_rds_symbols(h, rds_sym.data(), rds_count.data()), "fmd_get_rds_symbols");
        check(fmd_get_rds_bytes(h, rds_bytes.data(), bytes_cap, rds_bytes_count.data()), "fmd_get_rds_bytes");
        for (int c = 0; c < n_channels; c++) {
            auto a = GetAudioOut(c);
            obs_on_audio_block.Notify(a.data(), a.size(), rates.fs_audio);
            auto s = GetRDSPredSymbols(c);
            obs_on_rds_symbols.Notify(s.data(), s.size());
            if (rds_bytes_count[c] > 0) obs_on_rds_bytes.Notify(rds_bytes.data() + (size_t)c * bytes_cap, (size_t)rds_bytes_count[c]);
        }
    }
};

// reference App (src/app.h:19-47, app.cpp:39-65): accumulate arbitrary-sized u8 IQ pieces into exactly one block
// (ReconstructionBuffer semantics, utility/reconstruction_buffer.h:16-26), then run the demodulator on it.
class App_GPU {
    int block_size;
    std::vector<uint8_t> buf;   // [block_size][2]
    size_t length = 0;          // samples filled
    Broadcast_FM_Demod_GPU demod;
public:
    explicit App_GPU(int _block_size) : block_size(_block_size), buf((size_t)_block_size * 2), demod(_block_size, 1, 1024000) {}
    size_t Process(const uint8_t* iq, size_t n_samples) {
        size_t nb_read = 0;
        while (nb_read < n_samples) {
            const size_t want = (size_t)block_size - length;
            const size_t take = (n_samples - nb_read < want) ? (n_samples - nb_read) : want;
            std::memcpy(buf.data() + 2 * length, iq + 2 * nb_read, 2 * take);
            length += take;
            nb_read += take;
            if (length == (size_t)block_size) { demod.ProcessU8(buf.data(), (size_t)block_size); length = 0; }
        }
        return nb_read;
    }
    Broadcast_FM_Demod_GPU& GetFMDemod() { return demod; }
    auto& OnAudioBlock() { return demod.OnAudioOut(); }
    auto& On_RDS_Bytes() { return demod.On_RDS_Bytes(); }
};

}  // namespace fmd_host
