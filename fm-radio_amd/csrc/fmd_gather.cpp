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
#include "fmd_gather_plan.h"

namespace {

using fmd_gather_plan::kDepth;   // buffer sets on the collector: the block being consumed + two in flight
static_assert(fmd_gather_plan::kLoopbackRccl == FMD_GATHER_LOOPBACK_RCCL && fmd_gather_plan::kRotate == FMD_GATHER_ROTATE, "one set of flags");

struct RankCtx {
    int dev = 0;
    int comm_index = -1;         // index of the rank's device among the distinct devices (= its RCCL rank)
    bool local = false;          // lives on the collector's device: handed over by device-to-device copies
    fmd_handle h = nullptr;
    hipStream_t s = nullptr;
    void* pcm[kDepth] = {};      // FMD_GATHER_PCM16: conversion target, sent from here
    hipEvent_t ev_done[kDepth] = {};
    long k = 0;
    fmd_gather_plan::BlockBase blocks;   // the handle's block numbering against the gather's
};

}  // namespace

struct fmd_gather_s {
    fmd_gather_config cfg{};
    fmd_gather_plan::Plan plan;                  // who collects what, which shards cross RCCL (fmd_gather_plan.h: tested without a GPU)
    std::unique_ptr<fmd_gather_plan::Sync> sync; // host-side hand-shake of the rank threads and the collecting thread
    std::vector<std::unique_ptr<RankCtx>> r;
    std::vector<ncclComm_t> comms;
    std::vector<std::unique_ptr<std::mutex>> comm_mu;
    int C_local = 0, n_audio = 0, cap = 0;
    size_t audio_bytes = 0, bytes_bytes = 0, counts_bytes = 0;    // per rank and block
    // collector buffers: one set of kDepth per collecting device (one device unless FMD_GATHER_ROTATE), [device index][slot]
    std::vector<std::vector<void*>> out_audio;
    std::vector<std::vector<uint8_t*>> out_bytes;
    std::vector<std::vector<int*>> out_counts;
    std::vector<std::vector<hipEvent_t>> ev_recv;
    long waited = 0;
    std::mutex err_mu;                           // every rank thread and the collector's write `err`
    std::string err;
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
    { const std::string why = g->plan.init(cfg->n_ranks, cfg->devices, cfg->root, cfg->flags); if (!why.empty()) return bail(gfail(g, FMD_ERR_ARG, "%s", why.c_str())); }
    g->cfg.devices = g->plan.devices.data();
    g->sync.reset(new fmd_gather_plan::Sync(cfg->n_ranks));
    const fmd_gather_plan::Plan& P = g->plan;
    const int root_dev = P.devices[(size_t)cfg->root];
    // equal shards, one geometry
    fmd_rates r0{};
    std::vector<long> blk0((size_t)cfg->n_ranks, -1), epoch0((size_t)cfg->n_ranks, 0);
    for (int i = 0; i < cfg->n_ranks; i++) {
        if (!handles[i]) return bail(gfail(g, FMD_ERR_ARG, "rank %d: null handle", i));
        fmd_rates ri{};
        const uint8_t* b = nullptr; const int* c = nullptr; int cap = 0;
        if (fmd_get_rates(handles[i], &ri) != FMD_OK || fmd_rds_bytes_dev(handles[i], &b, &c, &cap) != FMD_OK) return bail(gfail(g, FMD_ERR_ARG, "rank %d: handle refused", i));
        fmd_config ci{};
        if (fmd_get_config(handles[i], &ci) != FMD_OK || fmd_outputs_block(handles[i], &blk0[(size_t)i]) != FMD_OK || fmd_outputs_epoch(handles[i], &epoch0[(size_t)i]) != FMD_OK) return bail(gfail(g, FMD_ERR_ARG, "rank %d: handle refused", i));
        if (ci.device != cfg->devices[i]) return bail(gfail(g, FMD_ERR_ARG, "rank %d: handle lives on device %d, not %d", i, ci.device, cfg->devices[i]));
        const int C = ci.n_channels;
        if (i == 0) { r0 = ri; g->C_local = C; g->n_audio = ri.n_audio; g->cap = cap; }
        else if (ri.n_audio != r0.n_audio || ri.fs_baseband != r0.fs_baseband || C != g->C_local || cap != g->cap)
            return bail(gfail(g, FMD_ERR_ARG, "rank %d: shard differs from rank 0's (equal shards needed: pad the last one with idle stations)", i));
    }
    g->audio_bytes = (size_t)g->C_local * g->n_audio * 2 * (cfg->format == FMD_GATHER_PCM16 ? sizeof(int16_t) : sizeof(float));
    g->bytes_bytes = (size_t)g->C_local * g->cap;
    g->counts_bytes = (size_t)g->C_local * sizeof(int);
    for (int i = 0; i < cfg->n_ranks; i++) {
        std::unique_ptr<RankCtx> rc(new RankCtx());
        rc->dev = P.devices[(size_t)i];
        rc->h = handles[i];
        rc->local = rc->dev == root_dev;
        rc->comm_index = P.comm_index[(size_t)i];
        rc->blocks.start(blk0[(size_t)i], epoch0[(size_t)i]);           // (a handle that has run blocks before the gather exists: its numbering is ahead by that much)
        g->r.push_back(std::move(rc));
    }
    g->comms.assign(g->plan.uniq.size(), nullptr);
    {
        ncclResult_t e = ncclCommInitAll(g->comms.data(), (int)g->plan.uniq.size(), g->plan.uniq.data());
        if (e != ncclSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "ncclCommInitAll over %zu devices: %s", g->plan.uniq.size(), ncclGetErrorString(e)));
    }
    for (size_t u = 0; u < g->plan.uniq.size(); u++) g->comm_mu.emplace_back(new std::mutex());
    for (auto& rc : g->r) {
        if (hipSetDevice(rc->dev) != hipSuccess || hipStreamCreateWithFlags(&rc->s, hipStreamNonBlocking) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "stream on device %d", rc->dev));
        for (int d = 0; d < kDepth; d++) {
            if (hipEventCreateWithFlags(&rc->ev_done[d], hipEventDisableTiming) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "event"));
            if (cfg->format == FMD_GATHER_PCM16 && P.needs_pcm_staging((int)(&rc - &g->r[0])) && hipMalloc(&rc->pcm[d], g->audio_bytes) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "staging"));
        }
    }
    const size_t n_coll = P.coll_ranks.size();
    g->out_audio.assign(n_coll, std::vector<void*>(kDepth, nullptr));
    g->out_bytes.assign(n_coll, std::vector<uint8_t*>(kDepth, nullptr));
    g->out_counts.assign(n_coll, std::vector<int*>(kDepth, nullptr));
    g->ev_recv.assign(n_coll, std::vector<hipEvent_t>(kDepth, nullptr));
    for (size_t ci = 0; ci < n_coll; ci++) {
        if (hipSetDevice(g->r[(size_t)g->plan.coll_ranks[ci]]->dev) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "hipSetDevice"));
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
    // After fmd_gather_abort a receive whose sender failed before it posted its send never completes, and neither would a synchronise of
    // the stream it is queued on: the communicators are aborted FIRST (every rank thread has left the library by now: the caller joins
    // them before it destroys — multi_gpu_host.hpp does), which completes what is queued on them; only then are the streams drained.
    const bool aborted = g->sync && g->sync->is_aborted();
    // (each communicator's lock is taken around its abort: a caller that destroys while a rank thread is still inside ncclSend / ncclRecv —
    //  against the contract of fmdemod_gather.h — then waits for that call instead of freeing the communicator under it)
    if (aborted) for (size_t i = 0; i < g->comms.size(); i++) if (g->comms[i]) {
        std::unique_lock<std::mutex> lk;
        if (i < g->comm_mu.size() && g->comm_mu[i]) lk = std::unique_lock<std::mutex>(*g->comm_mu[i]);
        (void)ncclCommAbort(g->comms[i]); g->comms[i] = nullptr;
    }
    for (auto& rc : g->r) {
        (void)hipSetDevice(rc->dev);
        if (rc->s) { (void)hipStreamSynchronize(rc->s); (void)hipStreamDestroy(rc->s); }
        for (int d = 0; d < kDepth; d++) { if (rc->ev_done[d]) (void)hipEventDestroy(rc->ev_done[d]); if (rc->pcm[d]) (void)hipFree(rc->pcm[d]); }
    }
    for (ncclComm_t c : g->comms) if (c) (void)ncclCommDestroy(c);
    for (size_t ci = 0; ci < g->out_audio.size(); ci++) {
        (void)hipSetDevice(g->r[(size_t)g->plan.coll_ranks[ci]]->dev);
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
    const fmd_gather_plan::Plan& P = g->plan;
    const bool pcm = g->cfg.format == FMD_GATHER_PCM16;
    const long k = rc.k;
    const int slot = P.slot(k);
    const int root = P.collector_rank(k), ci = P.collector_index(k);      // this block's collector
    // the collector's buffer set of block k was block k - kDepth's: its views must have been given back
    if (!g->sync->begin_submit(k)) return gfail(g, FMD_ERR_STATE, "gather aborted");
    G_HIP(g, hipSetDevice(rc.dev));
    {   // the handle's device views must be THIS block's (a handle under fmd_set_output_lag shows the block before: not supported here)
        long blk = -1, epoch = 0;
        G_FMD(g, rc.h, fmd_outputs_block(rc.h, &blk));
        G_FMD(g, rc.h, fmd_outputs_epoch(rc.h, &epoch));
        const long rel = rc.blocks.relative(blk, epoch, k);
        if (rel != k) return gfail(g, FMD_ERR_ARG, "rank %d: the handle's outputs are its block %ld (%ld since fmd_gather_create), the gather expects block %ld: one new block per fmd_gather_submit — a repeated submit, a skipped block, or a handle under fmd_set_output_lag (whose views are the block before: not supported here)", rank, blk, rel, k);
    }
    // the block's outputs, behind its last stage, on the rank's gather stream
    const void* src_audio = nullptr;
    const bool via_rccl = P.via_rccl(rank, k);
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
    const std::vector<int> recvs = rank == root ? P.receives(k) : std::vector<int>();
    if (via_rccl || !recvs.empty()) {
        std::lock_guard<std::mutex> lk(*g->comm_mu[(size_t)rc.comm_index]);
        ncclComm_t comm = g->comms[(size_t)rc.comm_index];
        const int root_peer = g->r[(size_t)root]->comm_index;
        G_NCCL(g, ncclGroupStart());
        if (via_rccl) {
            G_NCCL(g, ncclSend(src_audio, g->audio_bytes, ncclUint8, root_peer, comm, rc.s));
            G_NCCL(g, ncclSend(d_bytes, g->bytes_bytes, ncclUint8, root_peer, comm, rc.s));
            G_NCCL(g, ncclSend(d_counts, g->counts_bytes, ncclUint8, root_peer, comm, rc.s));
        }
        for (int q : recvs) {   // the collector posts the receives of every shard that travels over RCCL (in rank order per peer)
            const RankCtx& rq = *g->r[(size_t)q];
            G_NCCL(g, ncclRecv(static_cast<char*>(g->out_audio[(size_t)ci][(size_t)slot]) + g->audio_bytes * q, g->audio_bytes, ncclUint8, rq.comm_index, comm, rc.s));
            G_NCCL(g, ncclRecv(g->out_bytes[(size_t)ci][(size_t)slot] + g->bytes_bytes * q, g->bytes_bytes, ncclUint8, rq.comm_index, comm, rc.s));
            G_NCCL(g, ncclRecv(g->out_counts[(size_t)ci][(size_t)slot] + (size_t)g->C_local * q, g->counts_bytes, ncclUint8, rq.comm_index, comm, rc.s));
        }
        G_NCCL(g, ncclGroupEnd());
    }
    // the library may reuse the block's buffers once everything queued on rc.s so far has read them
    G_FMD(g, rc.h, fmd_release_outputs(rc.h, rc.s));
    G_HIP(g, hipEventRecord(rank == root ? g->ev_recv[(size_t)ci][(size_t)slot] : rc.ev_done[slot], rc.s));
    if (rank == root) G_HIP(g, hipEventRecord(rc.ev_done[slot], rc.s));
    rc.k = k + 1;
    g->sync->end_submit(rank, k);
    return FMD_OK;
}

int fmd_gather_wait(fmd_gather g, const void** d_audio, const uint8_t** d_rds_bytes, const int** d_rds_counts, int* rds_cap) {
    if (!g) return FMD_ERR_ARG;
    const fmd_gather_plan::Plan& P = g->plan;
    const long w = g->waited;
    const int slot = P.slot(w);
    const int root = P.collector_rank(w), ci = P.collector_index(w);
    g->sync->begin_wait(w);                                  // the previous call's views are given back
    G_HIP(g, hipSetDevice(P.collector_device(w)));
    for (int q : P.polled(w)) {                              // (the other shards arrive through the collector's receives)
        RankCtx& rq = *g->r[(size_t)q];
        if (!g->sync->wait_submitted(q, w)) return gfail(g, FMD_ERR_STATE, "gather aborted");
        // (polled, not hipEventSynchronize: a receive whose sender has failed only completes once fmd_gather_destroy has aborted the communicators)
        for (;;) {
            const hipError_t qe = hipEventQuery(q == root ? g->ev_recv[(size_t)ci][(size_t)slot] : rq.ev_done[slot]);
            if (qe == hipSuccess) break;
            if (qe != hipErrorNotReady) return gfail(g, FMD_ERR_DEVICE, "hipEventQuery: %s", hipGetErrorString(qe));
            if (g->sync->is_aborted()) return gfail(g, FMD_ERR_STATE, "gather aborted");
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
    if (rank) *rank = g->plan.collector_rank(block);
    if (device) *device = g->plan.collector_device(block);
    return FMD_OK;
}

int fmd_gather_abort(fmd_gather g) {
    if (!g) return FMD_ERR_ARG;
    // Host side only: every waiter returns FMD_ERR_STATE from here on.  The communicators are aborted by fmd_gather_destroy, before it
    // drains the rank streams — ncclCommAbort frees a communicator, and rank threads may still be inside a call that uses one here.
    (void)g->sync->abort();
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
