// Multi-GPU C++ host of the batched demodulator (BASELINE configs[3]: stations sharded over the GPUs of one node).
//
// The reference wires ONE demodulator to an audio observer and an RDS byte chain (src/app.cpp:19-34).  `MultiGpuHost` is that for
// n_ranks x C_local stations: one `fmd_handle` and one host thread per rank (the C ABI's threading contract: one caller thread per
// handle), stations in contiguous ranges — rank r owns [r C_local, (r + 1) C_local) — and after every block the ranks' outputs
// (audio as f32 or 16-bit PCM frames, the Manchester decoder's byte buffers) are gathered onto the collecting rank's GPU over RCCL
// (include/fmdemod_gather.h).  Nothing else crosses GPUs.
//
//   caller thread                     rank threads (one per GPU)                        collector (caller thread again)
//   Submit(d_iq per rank) ---------->  fmd_submit_u8_dev / _cf32_dev on the rank's GPU
//                                      fmd_gather_submit: staging behind the outputs,
//                                      ncclSend (collector: ncclRecv from every GPU)
//   Collect(&views) <---------------------------------------------------------------   fmd_gather_wait: device views of the whole batch
//
// The IQ blocks are DEVICE pointers on each rank's GPU (the caller's rings live there: station_ring.hpp shows the PCIe side for one
// GPU); Submit never waits for a GPU.  Up to two blocks may be in flight beyond the one Collect has handed out.
#pragma once

#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "fmdemod.h"
#include "fmdemod_gather.h"

namespace fmd_host {

class MultiGpuHost {
public:
    struct Views {                       // device views on the block's collector GPU (`device`), valid until the next Collect()
        int device = 0;                  // HIP ordinal the views live on: the root's, or with FMD_GATHER_ROTATE the device of rank (root + block) % n_ranks
        const void* audio = nullptr;     // [n_ranks * C_local][n_audio][2] float or int16
        const uint8_t* rds_bytes = nullptr;   // [n_ranks * C_local][rds_cap]
        const int* rds_counts = nullptr;      // [n_ranks * C_local]
        int rds_cap = 0;
    };

    // devices: HIP ordinal per rank (may repeat: several handles on one GPU; only the collector's device may be shared)
    MultiGpuHost(const std::vector<int>& devices, int stations_per_rank, int block_size, int fs_baseband, unsigned demod_flags,
                 int gather_format = FMD_GATHER_PCM16, unsigned gather_flags = 0, int root = 0)
        : devs(devices), C(stations_per_rank), N(block_size), ranks(devices.size()) {
        if (devices.empty()) throw std::invalid_argument("MultiGpuHost: no devices");
        handles.assign(devices.size(), nullptr);
        for (size_t r = 0; r < devices.size(); r++) {
            fmd_config cfg{stations_per_rank, block_size, fs_baseband, devices[r], demod_flags};
            if (fmd_create(&cfg, &handles[r]) != FMD_OK) { const std::string e = fmd_last_error(nullptr); cleanup(); throw std::runtime_error("fmd_create on device " + std::to_string(devices[r]) + ": " + e); }
        }
        fmd_gather_config gc{(int)devices.size(), devs.data(), root, gather_format, gather_flags};
        if (fmd_gather_create(&gc, handles.data(), &gather) != FMD_OK) { const std::string e = fmd_gather_last_error(nullptr); cleanup(); throw std::runtime_error("fmd_gather_create: " + e); }
        fmd_get_rates(handles[0], &rates);
        for (size_t r = 0; r < devices.size(); r++) ranks[r].th = std::thread([this, r] { run((int)r); });
    }
    ~MultiGpuHost() {
        // blocks that were submitted and never collected are dropped: a rank thread waiting for the collector's buffers must not keep join() waiting
        bool pending = false;
        for (Rank& k : ranks) { std::lock_guard<std::mutex> lk(k.mu); k.stop = true; pending = pending || !k.q.empty() || !k.error.empty(); }
        if (gather && (pending || collected < submitted)) fmd_gather_abort(gather);
        for (Rank& k : ranks) k.cv.notify_all();
        for (Rank& k : ranks) if (k.th.joinable()) k.th.join();
        cleanup();
    }
    MultiGpuHost(const MultiGpuHost&) = delete;
    MultiGpuHost& operator=(const MultiGpuHost&) = delete;

    // One block on every GPU: d_iq[r] is rank r's [C_local][N][2] block on ITS device (u8 or cf32), in place when the call is made
    // and until the rank's front end has read it (fmd_wait_input on the rank's handle says when).  Returns at once.
    void SubmitU8(const std::vector<const uint8_t*>& d_iq) { push(d_iq.data(), true); }
    void SubmitCF32(const std::vector<const float*>& d_iq) { push(reinterpret_cast<const uint8_t* const*>(d_iq.data()), false); }

    // The oldest block not yet collected, from every rank: blocks until it has arrived on the collector's GPU.
    // A rank that failed (its error is what this throws) aborts the gather, so that this call does not wait for a block that never comes.
    Views Collect() {
        Views v;
        rethrow();
        const int rc = fmd_gather_wait(gather, &v.audio, &v.rds_bytes, &v.rds_counts, &v.rds_cap);
        rethrow();
        if (rc != FMD_OK) throw std::runtime_error(std::string("fmd_gather_wait: ") + fmd_gather_last_error(gather));
        fmd_gather_collector(gather, collected, nullptr, &v.device);
        collected++;
        return v;
    }

    int Ranks() const { return (int)handles.size(); }
    int StationsPerRank() const { return C; }
    const fmd_rates& Rates() const { return rates; }
    fmd_handle Handle(int rank) { return handles[(size_t)rank]; }
    size_t RemoteBytesPerBlock() const { return fmd_gather_remote_bytes_per_block(gather); }

private:
    struct Task { const uint8_t* iq; bool u8; };
    struct Rank {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Task> q;
        bool stop = false;
        std::string error;
    };

    void push(const uint8_t* const* d_iq, bool u8) {
        rethrow();
        submitted++;
        for (size_t r = 0; r < ranks.size(); r++) {
            { std::lock_guard<std::mutex> lk(ranks[r].mu); ranks[r].q.push_back(Task{d_iq[r], u8}); }
            ranks[r].cv.notify_one();
        }
    }
    void run(int r) {
        Rank& k = ranks[(size_t)r];
        (void)hipSetDevice(devs[(size_t)r]);
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(k.mu);
                k.cv.wait(lk, [&] { return k.stop || !k.q.empty(); });
                if (k.q.empty()) return;
                t = k.q.front(); k.q.pop_front();
            }
            { std::lock_guard<std::mutex> lk(k.mu); if (!k.error.empty() || k.stop) continue; }   // (keep draining: the caller sees the error at its next call)
            const int rc = t.u8 ? fmd_submit_u8_dev(handles[(size_t)r], t.iq, C, N, nullptr)
                                : fmd_submit_cf32_dev(handles[(size_t)r], reinterpret_cast<const float*>(t.iq), C, N, nullptr);
            // (a failed rank aborts the gather: the collector and the other ranks must not wait for its block)
            if (rc != FMD_OK) {
                const std::string e = "rank " + std::to_string(r) + ": fmd_submit: " + fmd_last_error(handles[(size_t)r]);
                note_first(e);
                { std::lock_guard<std::mutex> lk(k.mu); k.error = e; }
                fmd_gather_abort(gather);
                continue;
            }
            if (fmd_gather_submit(gather, r) != FMD_OK) {
                const std::string e = "rank " + std::to_string(r) + ": fmd_gather_submit: " + fmd_gather_last_error(gather);
                note_first(e);
                { std::lock_guard<std::mutex> lk(k.mu); if (!k.stop) k.error = e; }
                fmd_gather_abort(gather);
            }
        }
    }
    void rethrow() {
        // the FIRST failure is the cause: the other ranks' "gather aborted" errors follow from it
        { std::lock_guard<std::mutex> lk(first_mu); if (!first_error.empty()) throw std::runtime_error("MultiGpuHost: " + first_error); }
        for (Rank& k : ranks) { std::lock_guard<std::mutex> lk(k.mu); if (!k.error.empty()) throw std::runtime_error("MultiGpuHost: " + k.error); }
    }
    void note_first(const std::string& e) { std::lock_guard<std::mutex> lk(first_mu); if (first_error.empty()) first_error = e; }
    void cleanup() {
        if (gather) { fmd_gather_destroy(gather); gather = nullptr; }
        for (fmd_handle& h : handles) if (h) { fmd_destroy(h); h = nullptr; }
    }

    std::vector<int> devs;
    int C, N;
    std::vector<fmd_handle> handles;
    fmd_gather gather = nullptr;
    long submitted = 0, collected = 0;   // blocks (caller thread only)
    fmd_rates rates{};
    std::vector<Rank> ranks;
    std::mutex first_mu;
    std::string first_error;
};

}  // namespace fmd_host
