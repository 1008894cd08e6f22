// libfmdgather.so (include/fmdemod_gather.h): the per-block gather of the sharded demodulators' outputs — audio (f32 or 16-bit PCM)
// and the RDS byte buffers — to a collecting GPU, over RCCL point-to-point.  One process, one host thread per rank; see the header
// for the process model and why same-device ranks are handed over by a copy.  Links libfmdemod.so (the C ABI only) and librccl.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "fmdemod_gather.h"

namespace {

constexpr int kDepth = 3;        // buffer sets on the collector: the block being consumed + two in flight

struct RankCtx {
    int dev = 0;
    int comm_index = -1;         // index of the rank's device among the distinct devices (= its RCCL rank)
    bool local = false;          // lives on the collector's device: handed over by device-to-device copies
    fmd_handle h = nullptr;
    hipStream_t s = nullptr;
    void* pcm[kDepth] = {};      // FMD_GATHER_PCM16: conversion target, sent from here
    hipEvent_t ev_done[kDepth] = {};
    std::atomic<long> submitted{0};
    long k = 0;
};

}  // namespace

struct fmd_gather_s {
    fmd_gather_config cfg{};
    std::vector<int> devices;
    std::vector<std::unique_ptr<RankCtx>> r;
    std::vector<int> uniq;                       // distinct devices, RCCL rank order
    std::vector<ncclComm_t> comms;
    std::vector<std::unique_ptr<std::mutex>> comm_mu;
    int C_local = 0, n_audio = 0, cap = 0;
    size_t audio_bytes = 0, bytes_bytes = 0, counts_bytes = 0;    // per rank and block
    // collector buffers: one set of kDepth per collecting device (one device unless FMD_GATHER_ROTATE), [device index][slot]
    std::vector<std::vector<void*>> out_audio;
    std::vector<std::vector<uint8_t*>> out_bytes;
    std::vector<std::vector<int*>> out_counts;
    std::vector<std::vector<hipEvent_t>> ev_recv;
    std::vector<int> coll_ranks;                 // the ranks that collect, in rotation order (one entry without FMD_GATHER_ROTATE)
    std::atomic<long> released{0};               // blocks whose views the collector has given back (= fmd_gather_wait calls begun)
    std::atomic<bool> aborted{false};
    long waited = 0;
    std::mutex err_mu;                           // every rank thread and the collector's write `err`
    std::string err;

    int collector_rank(long k) const { return coll_ranks[(size_t)(k % (long)coll_ranks.size())]; }
    int collector_index(long k) const { return (int)(k % (long)coll_ranks.size()); }
};

namespace {

thread_local std::string g_gather_create_error;

int gfail(fmd_gather g, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g) { std::lock_guard<std::mutex> lk(g->err_mu); g->err = buf; } else g_gather_create_error = buf;
    return code;
}

#define G_HIP(g, expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return gfail((g), FMD_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define G_NCCL(g, expr)                                                                                   \
    do {                                                                                                  \
        ncclResult_t e_ = (expr);                                                                         \
        if (e_ != ncclSuccess) return gfail((g), FMD_ERR_DEVICE, "%s: %s", #expr, ncclGetErrorString(e_)); \
    } while (0)
#define G_FMD(g, h, expr)                                                                      \
    do {                                                                                       \
        int rc_ = (expr);                                                                      \
        if (rc_ != FMD_OK) return gfail((g), rc_, "%s: %s", #expr, fmd_last_error(h));         \
    } while (0)

}  // namespace

extern "C" {

int fmd_gather_create(const fmd_gather_config* cfg, const fmd_handle* handles, fmd_gather* out) {
    if (!out) return FMD_ERR_ARG;
    *out = nullptr;
    if (!cfg || !handles || cfg->n_ranks <= 0 || !cfg->devices || cfg->root < 0 || cfg->root >= cfg->n_ranks ||
        (cfg->format != FMD_GATHER_F32 && cfg->format != FMD_GATHER_PCM16) || (cfg->flags & ~(FMD_GATHER_LOOPBACK_RCCL | FMD_GATHER_ROTATE)))
        return gfail(nullptr, FMD_ERR_ARG, "bad gather configuration");
    fmd_gather g = new (std::nothrow) fmd_gather_s();
    if (!g) return FMD_ERR_ARG;
    auto bail = [&](int rc) { g_gather_create_error = g->err; fmd_gather_destroy(g); return rc; };
    g->cfg = *cfg;
    g->devices.assign(cfg->devices, cfg->devices + cfg->n_ranks);
    g->cfg.devices = g->devices.data();
    const int root_dev = g->devices[(size_t)cfg->root];
    const bool loopback = (cfg->flags & FMD_GATHER_LOOPBACK_RCCL) != 0;
    // equal shards, one geometry
    fmd_rates r0{};
    for (int i = 0; i < cfg->n_ranks; i++) {
        if (!handles[i]) return bail(gfail(g, FMD_ERR_ARG, "rank %d: null handle", i));
        fmd_rates ri{};
        const uint8_t* b = nullptr; const int* c = nullptr; int cap = 0;
        if (fmd_get_rates(handles[i], &ri) != FMD_OK || fmd_rds_bytes_dev(handles[i], &b, &c, &cap) != FMD_OK) return bail(gfail(g, FMD_ERR_ARG, "rank %d: handle refused", i));
        fmd_config ci{};
        if (fmd_get_config(handles[i], &ci) != FMD_OK) return bail(gfail(g, FMD_ERR_ARG, "rank %d: handle refused", i));
        if (ci.device != cfg->devices[i]) return bail(gfail(g, FMD_ERR_ARG, "rank %d: handle lives on device %d, not %d", i, ci.device, cfg->devices[i]));
        const int C = ci.n_channels;
        if (i == 0) { r0 = ri; g->C_local = C; g->n_audio = ri.n_audio; g->cap = cap; }
        else if (ri.n_audio != r0.n_audio || ri.fs_baseband != r0.fs_baseband || C != g->C_local || cap != g->cap)
            return bail(gfail(g, FMD_ERR_ARG, "rank %d: shard differs from rank 0's (equal shards needed: pad the last one with idle stations)", i));
    }
    g->audio_bytes = (size_t)g->C_local * g->n_audio * 2 * (cfg->format == FMD_GATHER_PCM16 ? sizeof(int16_t) : sizeof(float));
    g->bytes_bytes = (size_t)g->C_local * g->cap;
    g->counts_bytes = (size_t)g->C_local * sizeof(int);
    // ranks, distinct devices
    int on_root_dev = 0;
    for (int i = 0; i < cfg->n_ranks; i++) {
        std::unique_ptr<RankCtx> rc(new RankCtx());
        rc->dev = g->devices[(size_t)i];
        rc->h = handles[i];
        rc->local = rc->dev == root_dev;
        on_root_dev += rc->local ? 1 : 0;
        int idx = -1;
        for (size_t u = 0; u < g->uniq.size(); u++) if (g->uniq[u] == rc->dev) idx = (int)u;
        if (idx < 0) { idx = (int)g->uniq.size(); g->uniq.push_back(rc->dev); }
        else if (!rc->local) return bail(gfail(g, FMD_ERR_ARG, "rank %d: a second rank on device %d, which is not the collector's (sends of two ranks through one communicator cannot be ordered)", i, rc->dev));
        rc->comm_index = idx;
        g->r.push_back(std::move(rc));
    }
    if (loopback && on_root_dev != 1) return bail(gfail(g, FMD_ERR_ARG, "FMD_GATHER_LOOPBACK_RCCL needs the collector alone on its device"));
    // who collects: the root alone, or every rank in turn (one rank per device then: a rank is "local" to one collector only)
    const bool rotate = (cfg->flags & FMD_GATHER_ROTATE) != 0 && g->uniq.size() > 1;
    if (rotate && (int)g->uniq.size() != cfg->n_ranks) return bail(gfail(g, FMD_ERR_ARG, "FMD_GATHER_ROTATE needs one rank per device (%d ranks on %zu devices)", cfg->n_ranks, g->uniq.size()));
    for (int i = 0; i < (rotate ? cfg->n_ranks : 1); i++) g->coll_ranks.push_back((cfg->root + i) % cfg->n_ranks);
    g->comms.assign(g->uniq.size(), nullptr);
    {
        ncclResult_t e = ncclCommInitAll(g->comms.data(), (int)g->uniq.size(), g->uniq.data());
        if (e != ncclSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "ncclCommInitAll over %zu devices: %s", g->uniq.size(), ncclGetErrorString(e)));
    }
    for (size_t u = 0; u < g->uniq.size(); u++) g->comm_mu.emplace_back(new std::mutex());
    for (auto& rc : g->r) {
        if (hipSetDevice(rc->dev) != hipSuccess || hipStreamCreateWithFlags(&rc->s, hipStreamNonBlocking) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "stream on device %d", rc->dev));
        for (int d = 0; d < kDepth; d++) {
            if (hipEventCreateWithFlags(&rc->ev_done[d], hipEventDisableTiming) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "event"));
            if (cfg->format == FMD_GATHER_PCM16 && (!rc->local || loopback || rotate) && hipMalloc(&rc->pcm[d], g->audio_bytes) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "staging"));
        }
    }
    const size_t n_coll = g->coll_ranks.size();
    g->out_audio.assign(n_coll, std::vector<void*>(kDepth, nullptr));
    g->out_bytes.assign(n_coll, std::vector<uint8_t*>(kDepth, nullptr));
    g->out_counts.assign(n_coll, std::vector<int*>(kDepth, nullptr));
    g->ev_recv.assign(n_coll, std::vector<hipEvent_t>(kDepth, nullptr));
    for (size_t ci = 0; ci < n_coll; ci++) {
        if (hipSetDevice(g->r[(size_t)g->coll_ranks[ci]]->dev) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "hipSetDevice"));
        for (int d = 0; d < kDepth; d++) {
            if (hipMalloc(&g->out_audio[ci][d], g->audio_bytes * cfg->n_ranks) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&g->out_bytes[ci][d]), g->bytes_bytes * cfg->n_ranks) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&g->out_counts[ci][d]), g->counts_bytes * cfg->n_ranks) != hipSuccess || hipEventCreateWithFlags(&g->ev_recv[ci][d], hipEventDisableTiming) != hipSuccess)
                return bail(gfail(g, FMD_ERR_DEVICE, "collector buffers"));
        }
    }
    *out = g;
    return FMD_OK;
}

int fmd_gather_destroy(fmd_gather g) {
    if (!g) return FMD_ERR_ARG;
    for (auto& rc : g->r) {
        (void)hipSetDevice(rc->dev);
        if (rc->s) { (void)hipStreamSynchronize(rc->s); (void)hipStreamDestroy(rc->s); }
        for (int d = 0; d < kDepth; d++) { if (rc->ev_done[d]) (void)hipEventDestroy(rc->ev_done[d]); if (rc->pcm[d]) (void)hipFree(rc->pcm[d]); }
    }
    for (ncclComm_t c : g->comms) if (c) (void)(g->aborted.load() ? ncclCommAbort(c) : ncclCommDestroy(c));
    for (size_t ci = 0; ci < g->out_audio.size(); ci++) {
        (void)hipSetDevice(g->r[(size_t)g->coll_ranks[ci]]->dev);
        for (int d = 0; d < kDepth; d++) {
            if (g->out_audio[ci][d]) (void)hipFree(g->out_audio[ci][d]);
            if (g->out_bytes[ci][d]) (void)hipFree(g->out_bytes[ci][d]);
            if (g->out_counts[ci][d]) (void)hipFree(g->out_counts[ci][d]);
            if (g->ev_recv[ci][d]) (void)hipEventDestroy(g->ev_recv[ci][d]);
        }
    }
    delete g;
    return FMD_OK;
}

int fmd_gather_submit(fmd_gather g, int rank) {
    if (!g || rank < 0 || rank >= g->cfg.n_ranks) return FMD_ERR_ARG;
    RankCtx& rc = *g->r[(size_t)rank];
    const bool loopback = (g->cfg.flags & FMD_GATHER_LOOPBACK_RCCL) != 0, pcm = g->cfg.format == FMD_GATHER_PCM16;
    const long k = rc.k;
    const int slot = (int)(k % kDepth);
    const int root = g->collector_rank(k), ci = g->collector_index(k);      // this block's collector
    const bool same_dev = rc.dev == g->r[(size_t)root]->dev;
    // the collector's buffer set of block k was block k - kDepth's: its views must have been given back
    while (k - kDepth + 1 > g->released.load(std::memory_order_acquire)) {
        if (g->aborted.load(std::memory_order_acquire)) return gfail(g, FMD_ERR_STATE, "gather aborted");
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    if (g->aborted.load(std::memory_order_acquire)) return gfail(g, FMD_ERR_STATE, "gather aborted");
    G_HIP(g, hipSetDevice(rc.dev));
    {   // the handle's device views must be THIS block's (a handle under fmd_set_output_lag shows the block before: not supported here)
        long blk = -1;
        G_FMD(g, rc.h, fmd_outputs_block(rc.h, &blk));
        if (blk != k) return gfail(g, FMD_ERR_ARG, "rank %d: the handle's outputs are block %ld, the gather expects block %ld (submit one block per fmd_gather_submit; handles under fmd_set_output_lag are not supported)", rank, blk, k);
    }
    // the block's outputs, behind its last stage, on the rank's gather stream
    const void* src_audio = nullptr;
    const bool via_rccl = !same_dev || loopback;
    if (pcm && via_rccl) {
        G_FMD(g, rc.h, fmd_audio_pcm16_dev(rc.h, static_cast<int16_t*>(rc.pcm[slot]), rc.s));     // (waits for the outputs on rc.s)
        src_audio = rc.pcm[slot];
    } else if (pcm) {   // a copy hand-over: convert straight into the collector's buffer
        G_FMD(g, rc.h, fmd_audio_pcm16_dev(rc.h, reinterpret_cast<int16_t*>(static_cast<char*>(g->out_audio[(size_t)ci][(size_t)slot]) + g->audio_bytes * rank), rc.s));
    } else {
        G_FMD(g, rc.h, fmd_wait_outputs(rc.h, rc.s));
        const float* p = nullptr;
        G_FMD(g, rc.h, fmd_audio_dev(rc.h, &p));
        src_audio = p;
    }
    if (pcm) G_FMD(g, rc.h, fmd_wait_outputs(rc.h, rc.s));   // (already waited for by the conversion; keeps the RDS views ordered too)
    const uint8_t* d_bytes = nullptr; const int* d_counts = nullptr; int cap = 0;
    G_FMD(g, rc.h, fmd_rds_bytes_dev(rc.h, &d_bytes, &d_counts, &cap));
    char* o_audio = static_cast<char*>(g->out_audio[(size_t)ci][(size_t)slot]) + g->audio_bytes * rank;
    uint8_t* o_bytes = g->out_bytes[(size_t)ci][(size_t)slot] + g->bytes_bytes * rank;
    int* o_counts = g->out_counts[(size_t)ci][(size_t)slot] + (size_t)g->C_local * rank;
    if (!via_rccl) {
        if (src_audio) G_HIP(g, hipMemcpyAsync(o_audio, src_audio, g->audio_bytes, hipMemcpyDeviceToDevice, rc.s));
        G_HIP(g, hipMemcpyAsync(o_bytes, d_bytes, g->bytes_bytes, hipMemcpyDeviceToDevice, rc.s));
        G_HIP(g, hipMemcpyAsync(o_counts, d_counts, g->counts_bytes, hipMemcpyDeviceToDevice, rc.s));
    }
    if (via_rccl || rank == root) {
        std::lock_guard<std::mutex> lk(*g->comm_mu[(size_t)rc.comm_index]);
        ncclComm_t comm = g->comms[(size_t)rc.comm_index];
        const int root_peer = g->r[(size_t)root]->comm_index;
        G_NCCL(g, ncclGroupStart());
        if (via_rccl) {
            G_NCCL(g, ncclSend(src_audio, g->audio_bytes, ncclUint8, root_peer, comm, rc.s));
            G_NCCL(g, ncclSend(d_bytes, g->bytes_bytes, ncclUint8, root_peer, comm, rc.s));
            G_NCCL(g, ncclSend(d_counts, g->counts_bytes, ncclUint8, root_peer, comm, rc.s));
        }
        if (rank == root) {   // the collector posts the receives of every shard that travels over RCCL (in rank order per peer)
            for (int q = 0; q < g->cfg.n_ranks; q++) {
                const RankCtx& rq = *g->r[(size_t)q];
                if (rq.dev == rc.dev && !loopback) continue;
                G_NCCL(g, ncclRecv(static_cast<char*>(g->out_audio[(size_t)ci][(size_t)slot]) + g->audio_bytes * q, g->audio_bytes, ncclUint8, rq.comm_index, comm, rc.s));
                G_NCCL(g, ncclRecv(g->out_bytes[(size_t)ci][(size_t)slot] + g->bytes_bytes * q, g->bytes_bytes, ncclUint8, rq.comm_index, comm, rc.s));
                G_NCCL(g, ncclRecv(g->out_counts[(size_t)ci][(size_t)slot] + (size_t)g->C_local * q, g->counts_bytes, ncclUint8, rq.comm_index, comm, rc.s));
            }
        }
        G_NCCL(g, ncclGroupEnd());
    }
    // the library may reuse the block's buffers once everything queued on rc.s so far has read them
    G_FMD(g, rc.h, fmd_release_outputs(rc.h, rc.s));
    G_HIP(g, hipEventRecord(rank == root ? g->ev_recv[(size_t)ci][(size_t)slot] : rc.ev_done[slot], rc.s));
    if (rank == root) G_HIP(g, hipEventRecord(rc.ev_done[slot], rc.s));
    rc.k = k + 1;
    rc.submitted.store(k + 1, std::memory_order_release);
    return FMD_OK;
}

int fmd_gather_wait(fmd_gather g, const void** d_audio, const uint8_t** d_rds_bytes, const int** d_rds_counts, int* rds_cap) {
    if (!g) return FMD_ERR_ARG;
    const long w = g->waited;
    const int slot = (int)(w % kDepth);
    const int root = g->collector_rank(w), ci = g->collector_index(w);
    const int root_dev = g->r[(size_t)root]->dev;
    g->released.store(w, std::memory_order_release);        // the previous call's views are given back
    G_HIP(g, hipSetDevice(root_dev));
    for (int q = 0; q < g->cfg.n_ranks; q++) {
        RankCtx& rq = *g->r[(size_t)q];
        const bool copies = rq.dev == root_dev && !(g->cfg.flags & FMD_GATHER_LOOPBACK_RCCL);
        if (!copies && q != root) continue;                  // its shard arrives through the collector's receives
        while (rq.submitted.load(std::memory_order_acquire) <= w) {
            if (g->aborted.load(std::memory_order_acquire)) return gfail(g, FMD_ERR_STATE, "gather aborted");
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        // (polled, not hipEventSynchronize: a receive whose sender has failed only completes once fmd_gather_abort has aborted the communicators)
        for (;;) {
            const hipError_t qe = hipEventQuery(q == root ? g->ev_recv[(size_t)ci][(size_t)slot] : rq.ev_done[slot]);
            if (qe == hipSuccess) break;
            if (qe != hipErrorNotReady) return gfail(g, FMD_ERR_DEVICE, "hipEventQuery: %s", hipGetErrorString(qe));
            if (g->aborted.load(std::memory_order_acquire)) return gfail(g, FMD_ERR_STATE, "gather aborted");
            std::this_thread::sleep_for(std::chrono::microseconds(10));
        }
    }
    g->waited = w + 1;
    if (d_audio) *d_audio = g->out_audio[(size_t)ci][(size_t)slot];
    if (d_rds_bytes) *d_rds_bytes = g->out_bytes[(size_t)ci][(size_t)slot];
    if (d_rds_counts) *d_rds_counts = g->out_counts[(size_t)ci][(size_t)slot];
    if (rds_cap) *rds_cap = g->cap;
    return FMD_OK;
}

int fmd_gather_collector(fmd_gather g, long block, int* rank, int* device) {
    if (!g || block < 0) return FMD_ERR_ARG;
    const int r = g->collector_rank(block);
    if (rank) *rank = r;
    if (device) *device = g->r[(size_t)r]->dev;
    return FMD_OK;
}

int fmd_gather_abort(fmd_gather g) {
    if (!g) return FMD_ERR_ARG;
    if (g->aborted.exchange(true)) return FMD_OK;
    // (the communicators are aborted in fmd_gather_destroy: ncclCommAbort frees them, and rank threads may still be inside a call that uses one)
    return FMD_OK;
}

size_t fmd_gather_remote_bytes_per_block(fmd_gather g) {
    if (!g) return 0;
    size_t n = 0;
    const int root_dev = g->r[(size_t)g->cfg.root]->dev;
    for (auto& rc : g->r) if (rc->dev != root_dev) n += g->audio_bytes + g->bytes_bytes + g->counts_bytes;     // (the same for every collector of a rotation)
    return n;
}

const char* fmd_gather_last_error(fmd_gather g) {
    if (!g) return g_gather_create_error.c_str();
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g->err_mu);
    copy = g->err;
    return copy.c_str();
}

}  // extern "C"
