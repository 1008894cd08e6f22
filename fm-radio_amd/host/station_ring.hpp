// Multi-station host side of the MI355X demodulator: C++ host code that owns the IQ ring buffers.
//
// The reference runs ONE station per `App` (src/app.cpp:39-65): the device thread hands `App::Process` arbitrary-sized pieces
// of u8 IQ, a `ReconstructionBuffer` (src/utility/reconstruction_buffer.h:16-26) accumulates them until exactly one block is
// full, the block is converted and demodulated, observers fire.  `StationRing` is that, for C stations feeding one batched
// GPU demodulator:
//
//   producers (any threads, one per station at a time)        owner thread                       GPU
//   Push(station, iq, n) -> memcpy into the station's row     Poll(): block complete in every    H2D copy   (copy-in stream)
//   of a PINNED staging block [C][N][2] u8; a station may     station -> hipMemcpyAsync, then    fmd_submit_u8_dev (library streams)
//   run up to `depth - 1` blocks ahead of the slowest         fmd_submit_u8_dev, then async      D2H audio / RDS bytes (copy-out stream)
//   (ConsumeBuffer semantics: returns what it took)           D2H of the outputs; finished
//                                                             blocks -> observers, per station
//
// `depth` staging blocks rotate, so the PCIe copy of block k+1 and the producers' filling of block k+2 overlap the GPU's work
// on block k; nothing in the steady state blocks the host (`Poll` only queries events).  Outputs are fetched from the
// library's device views (fmd_audio_dev, fmd_rds_bytes_dev) by asynchronous copies, and `fmd_release_outputs` tells the
// library when those copies are done with a slot.
//
// Threading: `Push` is safe from several threads as long as no two threads push the SAME station concurrently (a station's
// fill state is touched by its producer only; block completion is an atomic counter).  Everything else — construction, Poll,
// Flush, observers — belongs to one owner thread, like the reference's single caller thread.  Observers fire on that thread,
// block after block, station 0 .. C-1 within a block.
#pragma once

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "fmdemod.h"

namespace fmd_host {

#ifndef FMD_HOST_FRAME_DEFINED
#define FMD_HOST_FRAME_DEFINED
struct Frame { float channels[2]; };  // reference src/audio/frame.h:6-8
#endif

class StationRing {
public:
    using AudioObserver = std::function<void(int station, const Frame* frames, size_t n, int Fs)>;   // reference OnAudioBlock (app.h)
    using BytesObserver = std::function<void(int station, const uint8_t* bytes, size_t n)>;         // reference On_RDS_Bytes

    StationRing(int n_stations, int block_size, int fs_baseband = 1024000, unsigned flags = 0, int depth = 3)
        : C(n_stations), N(block_size), D(depth), fill(n_stations), slots(depth) {
        if (depth < 2) throw std::invalid_argument("StationRing: depth >= 2");
        fmd_config cfg{n_stations, block_size, fs_baseband, -1, flags};
        if (fmd_create(&cfg, &h) != FMD_OK) throw std::runtime_error(std::string("fmd_create: ") + fmd_last_error(nullptr));
        fmd_get_rates(h, &rates);
        const uint8_t* db = nullptr; const int* dc = nullptr;
        fmd_rds_bytes_dev(h, &db, &dc, &bytes_cap);
        hip(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking), "stream");
        hip(hipStreamCreateWithFlags(&s_proc, hipStreamNonBlocking), "stream");
        hip(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking), "stream");
        const size_t in_bytes = (size_t)C * N * 2, audio_floats = (size_t)C * rates.n_audio * 2;
        for (Slot& s : slots) {
            hip(hipHostMalloc(reinterpret_cast<void**>(&s.h_in), in_bytes, hipHostMallocDefault), "pinned input block");
            hip(hipMalloc(reinterpret_cast<void**>(&s.d_in), in_bytes), "device input block");
            hip(hipHostMalloc(reinterpret_cast<void**>(&s.h_audio), audio_floats * sizeof(float), hipHostMallocDefault), "pinned audio block");
            hip(hipHostMalloc(reinterpret_cast<void**>(&s.h_bytes), (size_t)C * bytes_cap, hipHostMallocDefault), "pinned RDS bytes");
            hip(hipHostMalloc(reinterpret_cast<void**>(&s.h_counts), (size_t)C * sizeof(int), hipHostMallocDefault), "pinned RDS counts");
            hip(hipEventCreateWithFlags(&s.ev_copied, hipEventDisableTiming), "event");
            hip(hipEventCreateWithFlags(&s.ev_consumed, hipEventDisableTiming), "event");
            hip(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming), "event");
            s.missing.store(C);
        }
        for (int i = 0; i < D; i++) slots[(size_t)i].block.store(i);      // staging block i starts out collecting block i
        for (Fill& f : fill) { f.block = 0; f.length = 0; }
    }
    ~StationRing() {
        if (h) { (void)hipDeviceSynchronize(); fmd_destroy(h); }
        for (Slot& s : slots) {
            if (s.h_in) (void)hipHostFree(s.h_in);
            if (s.d_in) (void)hipFree(s.d_in);
            if (s.h_audio) (void)hipHostFree(s.h_audio);
            if (s.h_bytes) (void)hipHostFree(s.h_bytes);
            if (s.h_counts) (void)hipHostFree(s.h_counts);
            for (hipEvent_t e : {s.ev_copied, s.ev_consumed, s.ev_out}) if (e) (void)hipEventDestroy(e);
        }
        for (hipStream_t st : {s_in, s_proc, s_out}) if (st) (void)hipStreamDestroy(st);
    }
    StationRing(const StationRing&) = delete;
    StationRing& operator=(const StationRing&) = delete;

    // ReconstructionBuffer::ConsumeBuffer for one station: appends up to n_samples u8 IQ samples, block after block, and returns
    // how many it took.  It takes fewer (possibly 0) only when the station is `depth - 1` blocks ahead of what the GPU side
    // has released: call Poll() on the owner thread and offer the rest again (reference App::Process loops the same way).
    size_t Push(int station, const uint8_t* iq, size_t n_samples) {
        Fill& f = fill[(size_t)station];
        size_t taken = 0;
        while (taken < n_samples) {
            Slot& s = slots[(size_t)(f.block % D)];
            if (s.block.load(std::memory_order_acquire) != f.block) break;      // that staging block still belongs to an older block
            const size_t want = (size_t)N - f.length, have = n_samples - taken;
            const size_t take = have < want ? have : want;
            std::memcpy(s.h_in + ((size_t)station * N + f.length) * 2, iq + 2 * taken, 2 * take);
            f.length += take;
            taken += take;
            if (f.length == (size_t)N) {
                f.length = 0;
                f.block++;
                s.missing.fetch_sub(1, std::memory_order_acq_rel);             // the owner submits the block when this reaches 0
            }
        }
        return taken;
    }

    // Owner thread, never blocks: submits every block that all stations have completed, fires the observers of every block whose
    // outputs have arrived, recycles staging blocks.  Returns the number of blocks delivered by this call.
    int Poll() {
        int delivered = 0;
        for (;;) {   // submit in block order
            Slot& s = slots[(size_t)(next_submit % D)];
            if (s.block.load(std::memory_order_relaxed) != next_submit || s.missing.load(std::memory_order_acquire) != 0 || s.submitted) break;
            submit(s);
            next_submit++;
        }
        for (;;) {   // deliver in block order
            if (next_deliver == next_submit) break;
            Slot& s = slots[(size_t)(next_deliver % D)];
            if (hipEventQuery(s.ev_out) != hipSuccess) break;
            deliver(s);
            // the staging block may now be refilled for block next_deliver + D (its H2D copy finished long before its outputs arrived)
            s.submitted = false;
            s.missing.store(C, std::memory_order_relaxed);
            s.block.store(next_deliver + D, std::memory_order_release);
            next_deliver++;
            delivered++;
        }
        return delivered;
    }

    // Owner thread: waits until every submitted block has been delivered (a trailing partial block stays pending, like the
    // reference's ReconstructionBuffer).
    void Flush() {
        Poll();
        while (next_deliver != next_submit) {
            hip(hipEventSynchronize(slots[(size_t)(next_deliver % D)].ev_out), "hipEventSynchronize");
            Poll();
        }
    }

    void OnAudioBlock(AudioObserver o) { on_audio.push_back(std::move(o)); }
    void On_RDS_Bytes(BytesObserver o) { on_bytes.push_back(std::move(o)); }
    fmd_handle Handle() { return h; }
    const fmd_rates& Rates() const { return rates; }
    long BlocksDelivered() const { return next_deliver; }

private:
    struct Fill { long block; size_t length; char pad[48]; };    // one cache line per station: producers do not share lines
    struct Slot {
        uint8_t* h_in = nullptr; uint8_t* d_in = nullptr;
        float* h_audio = nullptr; uint8_t* h_bytes = nullptr; int* h_counts = nullptr;
        hipEvent_t ev_copied = nullptr, ev_consumed = nullptr, ev_out = nullptr;
        std::atomic<long> block{0};       // the block number this staging buffer currently collects
        std::atomic<int> missing{0};      // stations that have not completed it yet
        bool submitted = false;
        Slot() = default;
        Slot(const Slot&) {}              // vector(n) construction only
    };

    void hip(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string("StationRing: ") + what + ": " + hipGetErrorString(e)); }
    void check(int rc, const char* what) { if (rc != FMD_OK) throw std::runtime_error(std::string("StationRing: ") + what + ": " + fmd_last_error(h)); }

    void submit(Slot& s) {
        const size_t in_bytes = (size_t)C * N * 2;
        // the device input block was last read by the demodulator D blocks ago (ev_consumed, recorded behind fmd_wait_input)
        hip(hipStreamWaitEvent(s_in, s.ev_consumed, 0), "wait consumed");
        hip(hipMemcpyAsync(s.d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, s_in), "H2D");
        // submitted behind the copy on s_in; nothing is queued on s_in (the next block's copy does not wait for this block's front end)
        check(fmd_submit_u8_dev(h, s.d_in, C, N, s_in), "fmd_submit_u8_dev");
        check(fmd_wait_input(h, s_proc), "fmd_wait_input");
        hip(hipEventRecord(s.ev_consumed, s_proc), "record");   // fires once the library has consumed the input block
        // outputs: device views of the newest block -> pinned host memory, behind the block's last stage, on the copy-out stream
        check(fmd_wait_outputs(h, s_out), "fmd_wait_outputs");
        const float* d_audio = nullptr; const uint8_t* d_bytes = nullptr; const int* d_counts = nullptr; int cap = 0;
        check(fmd_audio_dev(h, &d_audio), "fmd_audio_dev");
        check(fmd_rds_bytes_dev(h, &d_bytes, &d_counts, &cap), "fmd_rds_bytes_dev");
        hip(hipMemcpyAsync(s.h_audio, d_audio, (size_t)C * rates.n_audio * 2 * sizeof(float), hipMemcpyDeviceToHost, s_out), "D2H audio");
        hip(hipMemcpyAsync(s.h_bytes, d_bytes, (size_t)C * cap, hipMemcpyDeviceToHost, s_out), "D2H RDS bytes");
        hip(hipMemcpyAsync(s.h_counts, d_counts, (size_t)C * sizeof(int), hipMemcpyDeviceToHost, s_out), "D2H RDS counts");
        check(fmd_release_outputs(h, s_out), "fmd_release_outputs");   // the library may reuse the slot once these copies are done
        hip(hipEventRecord(s.ev_out, s_out), "record");
        s.submitted = true;
    }

    void deliver(Slot& s) {
        for (int c = 0; c < C; c++) {
            const Frame* a = reinterpret_cast<const Frame*>(s.h_audio) + (size_t)c * rates.n_audio;
            for (auto& o : on_audio) o(c, a, (size_t)rates.n_audio, rates.fs_audio);
            const int nb = s.h_counts[c] < bytes_cap ? s.h_counts[c] : bytes_cap;
            if (nb > 0) for (auto& o : on_bytes) o(c, s.h_bytes + (size_t)c * bytes_cap, (size_t)nb);
        }
    }

    int C, N, D;
    fmd_handle h = nullptr;
    fmd_rates rates{};
    int bytes_cap = 0;
    hipStream_t s_in = nullptr, s_proc = nullptr, s_out = nullptr;
    std::vector<Fill> fill;
    std::vector<Slot> slots;
    long next_submit = 0, next_deliver = 0;
    std::vector<AudioObserver> on_audio;
    std::vector<BytesObserver> on_bytes;

};

}  // namespace fmd_host
