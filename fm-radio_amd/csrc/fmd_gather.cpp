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
    void* out_audio[kDepth] = {};
    uint8_t* out_bytes[kDepth] = {};
    int* out_counts[kDepth] = {};
    hipEvent_t ev_recv[kDepth] = {};
    std::atomic<long> released{0};               // blocks whose views the collector has given back (= fmd_gather_wait calls begun)
    long waited = 0;
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
    if (g) g->err = buf; else g_gather_create_error = buf;
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
        (cfg->format != FMD_GATHER_F32 && cfg->format != FMD_GATHER_PCM16) || (cfg->flags & ~FMD_GATHER_LOOPBACK_RCCL))
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
            if (cfg->format == FMD_GATHER_PCM16 && (!rc->local || loopback) && hipMalloc(&rc->pcm[d], g->audio_bytes) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "staging"));
        }
    }
    if (hipSetDevice(root_dev) != hipSuccess) return bail(gfail(g, FMD_ERR_DEVICE, "hipSetDevice"));
    for (int d = 0; d < kDepth; d++) {
        if (hipMalloc(&g->out_audio[d], g->audio_bytes * cfg->n_ranks) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&g->out_bytes[d]), g->bytes_bytes * cfg->n_ranks) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&g->out_counts[d]), g->counts_bytes * cfg->n_ranks) != hipSuccess || hipEventCreateWithFlags(&g->ev_recv[d], hipEventDisableTiming) != hipSuccess)
            return bail(gfail(g, FMD_ERR_DEVICE, "collector buffers"));
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
    for (ncclComm_t c : g->comms) if (c) (void)ncclCommDestroy(c);
    if (!g->devices.empty()) (void)hipSetDevice(g->devices[(size_t)g->cfg.root]);
    for (int d = 0; d < kDepth; d++) {
        if (g->out_audio[d]) (void)hipFree(g->out_audio[d]);
        if (g->out_bytes[d]) (void)hipFree(g->out_bytes[d]);
        if (g->out_counts[d]) (void)hipFree(g->out_counts[d]);
        if (g->ev_recv[d]) (void)hipEventDestroy(g->ev_recv[d]);
    }
    delete g;
    return FMD_OK;
}

int fmd_gather_submit(fmd_gather g, int rank) {
    if (!g || rank < 0 || rank >= g->cfg.n_ranks) return FMD_ERR_ARG;
    RankCtx& rc = *g->r[(size_t)rank];
    const int root = g->cfg.root;
    const bool loopback = (g->cfg.flags & FMD_GATHER_LOOPBACK_RCCL) != 0, pcm = g->cfg.format == FMD_GATHER_PCM16;
    const long k = rc.k;
    const int slot = (int)(k % kDepth);
    // the collector's buffer set of block k was block k - kDepth's: its views must have been given back
    while (k - kDepth + 1 > g->released.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(20));
    G_HIP(g, hipSetDevice(rc.dev));
    // the block's outputs, behind its last stage, on the rank's gather stream
    const void* src_audio = nullptr;
    const bool via_rccl = !rc.local || loopback;
    if (pcm && via_rccl) {
        G_FMD(g, rc.h, fmd_audio_pcm16_dev(rc.h, static_cast<int16_t*>(rc.pcm[slot]), rc.s));     // (waits for the outputs on rc.s)
        src_audio = rc.pcm[slot];
    } else if (pcm) {   // a copy hand-over: convert straight into the collector's buffer
        G_FMD(g, rc.h, fmd_audio_pcm16_dev(rc.h, reinterpret_cast<int16_t*>(static_cast<char*>(g->out_audio[slot]) + g->audio_bytes * rank), rc.s));
    } else {
        G_FMD(g, rc.h, fmd_wait_outputs(rc.h, rc.s));
        const float* p = nullptr;
        G_FMD(g, rc.h, fmd_audio_dev(rc.h, &p));
        src_audio = p;
    }
    if (pcm) G_FMD(g, rc.h, fmd_wait_outputs(rc.h, rc.s));   // (already waited for by the conversion; keeps the RDS views ordered too)
    const uint8_t* d_bytes = nullptr; const int* d_counts = nullptr; int cap = 0;
    G_FMD(g, rc.h, fmd_rds_bytes_dev(rc.h, &d_bytes, &d_counts, &cap));
    char* o_audio = static_cast<char*>(g->out_audio[slot]) + g->audio_bytes * rank;
    uint8_t* o_bytes = g->out_bytes[slot] + g->bytes_bytes * rank;
    int* o_counts = g->out_counts[slot] + (size_t)g->C_local * rank;
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
                if (rq.local && !loopback) continue;
                G_NCCL(g, ncclRecv(static_cast<char*>(g->out_audio[slot]) + g->audio_bytes * q, g->audio_bytes, ncclUint8, rq.comm_index, comm, rc.s));
                G_NCCL(g, ncclRecv(g->out_bytes[slot] + g->bytes_bytes * q, g->bytes_bytes, ncclUint8, rq.comm_index, comm, rc.s));
                G_NCCL(g, ncclRecv(g->out_counts[slot] + (size_t)g->C_local * q, g->counts_bytes, ncclUint8, rq.comm_index, comm, rc.s));
            }
        }
        G_NCCL(g, ncclGroupEnd());
    }
    // the library may reuse the block's buffers once everything queued on rc.s so far has read them
    G_FMD(g, rc.h, fmd_release_outputs(rc.h, rc.s));
    G_HIP(g, hipEventRecord(rank == root ? g->ev_recv[slot] : rc.ev_done[slot], rc.s));
    if (rank == root) G_HIP(g, hipEventRecord(rc.ev_done[slot], rc.s));
    rc.k = k + 1;
    rc.submitted.store(k + 1, std::memory_order_release);
    return FMD_OK;
}

int fmd_gather_wait(fmd_gather g, const void** d_audio, const uint8_t** d_rds_bytes, const int** d_rds_counts, int* rds_cap) {
    if (!g) return FMD_ERR_ARG;
    const long w = g->waited;
    const int slot = (int)(w % kDepth);
    g->released.store(w, std::memory_order_release);        // the previous call's views are given back
    G_HIP(g, hipSetDevice(g->devices[(size_t)g->cfg.root]));
    for (int q = 0; q < g->cfg.n_ranks; q++) {
        RankCtx& rq = *g->r[(size_t)q];
        const bool copies = rq.local && !(g->cfg.flags & FMD_GATHER_LOOPBACK_RCCL);
        if (!copies && q != g->cfg.root) continue;          // its shard arrives through the collector's receives
        while (rq.submitted.load(std::memory_order_acquire) <= w) std::this_thread::sleep_for(std::chrono::microseconds(20));
        G_HIP(g, hipEventSynchronize(q == g->cfg.root ? g->ev_recv[slot] : rq.ev_done[slot]));
    }
    g->waited = w + 1;
    if (d_audio) *d_audio = g->out_audio[slot];
    if (d_rds_bytes) *d_rds_bytes = g->out_bytes[slot];
    if (d_rds_counts) *d_rds_counts = g->out_counts[slot];
    if (rds_cap) *rds_cap = g->cap;
    return FMD_OK;
}

size_t fmd_gather_remote_bytes_per_block(fmd_gather g) {
    if (!g) return 0;
    size_t n = 0;
    for (auto& rc : g->r) if (!rc->local) n += g->audio_bytes + g->bytes_bytes + g->counts_bytes;
    return n;
}

const char* fmd_gather_last_error(fmd_gather g) { return g ? g->err.c_str() : g_gather_create_error.c_str(); }

}  // extern "C"
