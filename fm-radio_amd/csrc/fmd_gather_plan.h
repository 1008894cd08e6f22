// Bookkeeping of the per-block output gather (include/fmdemod_gather.h), free of HIP and RCCL so that it runs — and is tested — without a
// GPU (tests/cpp/gather_plan_main.cpp): who collects block k, which of the collector's buffer sets it lands in, which shards travel over
// RCCL and which are handed over by a copy, and the host-side hand-shake between the rank threads and the collecting thread (back-pressure
// on the collector's buffer sets, abort).  fmd_gather.cpp is this plus the device calls.
// Reference anchor: one demodulator wired to an audio observer AND an RDS byte chain per station, src/app.cpp:19-34 — here n_ranks shards of them.
#pragma once

#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace fmd_gather_plan {

constexpr int kDepth = 3;        // buffer sets per collecting device: the block being consumed + two in flight
constexpr unsigned kLoopbackRccl = 1u, kRotate = 2u;     // = FMD_GATHER_LOOPBACK_RCCL, FMD_GATHER_ROTATE (static_asserted in fmd_gather.cpp)

struct Plan {
    int n_ranks = 0, root = 0;
    unsigned flags = 0;
    std::vector<int> devices;        // [n_ranks] device ordinal of every rank
    std::vector<int> uniq;           // distinct devices, RCCL rank order
    std::vector<int> comm_index;     // [n_ranks] index of the rank's device in uniq (= its RCCL rank)
    std::vector<int> coll_ranks;     // the ranks that collect, in rotation order (one entry without kRotate)
    bool rotate = false, loopback = false;

    // "" or why the configuration is refused
    std::string init(int n_ranks_, const int* devices_, int root_, unsigned flags_) {
        n_ranks = n_ranks_; root = root_; flags = flags_;
        if (n_ranks <= 0 || !devices_ || root < 0 || root >= n_ranks || (flags & ~(kLoopbackRccl | kRotate))) return "bad gather configuration";
        devices.assign(devices_, devices_ + n_ranks);
        loopback = (flags & kLoopbackRccl) != 0;
        const int root_dev = devices[(size_t)root];
        int on_root_dev = 0;
        uniq.clear(); comm_index.assign((size_t)n_ranks, -1);
        for (int i = 0; i < n_ranks; i++) {
            const bool local = devices[(size_t)i] == root_dev;
            on_root_dev += local ? 1 : 0;
            int idx = -1;
            for (size_t u = 0; u < uniq.size(); u++) if (uniq[u] == devices[(size_t)i]) idx = (int)u;
            if (idx < 0) { idx = (int)uniq.size(); uniq.push_back(devices[(size_t)i]); }
            else if (!local) return "rank " + std::to_string(i) + ": a second rank on device " + std::to_string(devices[(size_t)i]) +
                                    ", which is not the collector's (sends of two ranks through one communicator cannot be ordered)";
            comm_index[(size_t)i] = idx;
        }
        if (loopback && on_root_dev != 1) return "FMD_GATHER_LOOPBACK_RCCL needs the collector alone on its device";
        // who collects: the root alone, or every rank in turn.  Rotation needs every rank to be "local" to exactly the collectors on its own
        // device: one rank per device (everything crosses RCCL), or every rank on ONE device (a one-GPU box: every hand-over is a copy).
        rotate = (flags & kRotate) != 0 && n_ranks > 1;
        if (rotate && (int)uniq.size() != n_ranks && uniq.size() != 1)
            return "FMD_GATHER_ROTATE needs one rank per device, or every rank on one device (" + std::to_string(n_ranks) + " ranks on " + std::to_string(uniq.size()) + " devices)";
        coll_ranks.clear();
        for (int i = 0; i < (rotate ? n_ranks : 1); i++) coll_ranks.push_back((root + i) % n_ranks);
        return "";
    }
    int collectors() const { return (int)coll_ranks.size(); }
    int collector_index(long k) const { return (int)(k % (long)coll_ranks.size()); }      // which of the collectors' buffer groups
    int collector_rank(long k) const { return coll_ranks[(size_t)collector_index(k)]; }
    int collector_device(long k) const { return devices[(size_t)collector_rank(k)]; }
    static int slot(long k) { return (int)(k % kDepth); }
    // rank's shard of block k goes through ncclSend / ncclRecv (otherwise: device-to-device copies into the collector's buffers)
    bool via_rccl(int rank, long k) const { return devices[(size_t)rank] != collector_device(k) || loopback; }
    // the collector of block k posts receives for these ranks, in this order
    std::vector<int> receives(long k) const {
        std::vector<int> q;
        for (int r = 0; r < n_ranks; r++) if (via_rccl(r, k)) q.push_back(r);
        return q;
    }
    // whose completion events fmd_gather_wait polls for block k: the collector's (its receives cover every RCCL shard) and every copying rank's
    std::vector<int> polled(long k) const {
        std::vector<int> q;
        for (int r = 0; r < n_ranks; r++) if (r == collector_rank(k) || !via_rccl(r, k)) q.push_back(r);
        return q;
    }
    // needs staging of its own for 16-bit PCM (the conversion cannot go straight into a buffer on another device / through RCCL)
    bool needs_pcm_staging(int rank) const {
        for (long k = 0; k < (long)coll_ranks.size(); k++) if (via_rccl(rank, k)) return true;
        return false;
    }
};

// Host-side hand-shake.  Rank threads: begin_submit(k) (waits while the collector's buffer set of block k still holds block k - kDepth's
// views), ..., end_submit(rank, k).  Collecting thread: begin_wait(w) gives the previous views back, wait_submitted(rank, w).  abort()
// makes every waiter return false, now and from here on.
struct Sync {
    std::atomic<long> released{0};               // blocks whose views the collector has given back (= waits begun)
    std::atomic<bool> aborted{false};
    std::vector<std::unique_ptr<std::atomic<long>>> submitted;
    explicit Sync(int n_ranks) { for (int i = 0; i < n_ranks; i++) submitted.emplace_back(new std::atomic<long>(0)); }
    bool is_aborted() const { return aborted.load(std::memory_order_acquire); }
    bool abort() { return aborted.exchange(true); }      // true: was aborted already
    bool begin_submit(long k) const {
        while (k - kDepth + 1 > released.load(std::memory_order_acquire)) {
            if (is_aborted()) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        return !is_aborted();
    }
    void end_submit(int rank, long k) { submitted[(size_t)rank]->store(k + 1, std::memory_order_release); }
    void begin_wait(long w) { released.store(w, std::memory_order_release); }
    bool wait_submitted(int rank, long w) const {
        while (submitted[(size_t)rank]->load(std::memory_order_acquire) <= w) {
            if (is_aborted()) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        return true;
    }
};

// A handle's block numbering against the gather's: fmd_outputs_block counts since fmd_create / fmd_reset, the gather since
// fmd_gather_create.  base = the handle's block count when the gather was created.  The numbering restarts ONLY when the handle says so
// (fmd_outputs_epoch changes: fmd_reset), and then the first block after the restart — the handle's block 0 — is the gather's next one.
// Returns the gather-relative block of handle block `blk`; anything but expect_k is the caller's error to report: a second submit without a
// new block (blk == last), a skipped block, a handle under fmd_set_output_lag (its views are the block before: -1, then k - 1).
// (Round 5 inferred a restart from blk <= last, which also "re-based" exactly those repeated and lagged submits — ADVICE r5.)
struct BlockBase {
    long base = 0, last = -1, epoch = 0;
    void start(long outputs_block_at_create, long epoch_at_create = 0) { base = outputs_block_at_create + 1; last = outputs_block_at_create; epoch = epoch_at_create; }
    long relative(long blk, long epoch_now, long expect_k) {
        if (epoch_now != epoch) { epoch = epoch_now; base = -expect_k; }      // restarted: the handle's block 0 is the gather's block expect_k
        last = blk;
        return blk - base;
    }
};

}  // namespace fmd_gather_plan
