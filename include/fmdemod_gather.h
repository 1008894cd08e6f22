/* fmdemod_gather.h — C ABI of libfmdgather.so: the multi-GPU side of the batched demodulator (BASELINE configs[3]).
 *
 * Stations shard over the GPUs of a node in contiguous ranges (rank r owns stations [r * C_local, (r + 1) * C_local)); nothing is
 * exchanged while demodulating.  The one exchange is this gather of every block's OUTPUTS to a collecting rank, over RCCL
 * (point-to-point ncclSend / ncclRecv on the direct xGMI link between the two GPUs):
 *   - the audio block, as f32 frames or as the 16-bit PCM frames the reference's scraper writes (fmd_audio_pcm16_dev), and
 *   - the per-station RDS byte buffers of the on-GPU Manchester decoder with their counts
 * — what the reference hands its two observers per station, OnAudioOut() and the RDS byte chain (src/app.cpp:19-34).
 *
 * Process model: ONE process, one host thread per GPU (each owns its fmd_handle, as fmdemod.h's threading contract asks) —
 * the C++ host fm-radio_amd/host/multi_gpu_host.hpp is that.  RCCL communicators are created over the DISTINCT devices of the rank
 * list (ncclCommInitAll).  Ranks that share the collector's device (more handles than GPUs; the one-GPU test box) hand their
 * shard over by a device-to-device copy instead: RCCL refuses two ranks on one device (ncclInvalidUsage, tools/rccl_probe.cpp).
 *
 * (The Python path of bench.py --gpus N — one process per GPU under torch.distributed, fm-radio_amd/sharding.py — is the other
 * way to run the same partition; its collective is torch's RCCL binding.)
 */
#ifndef FMDEMOD_GATHER_H
#define FMDEMOD_GATHER_H

#include "fmdemod.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fmd_gather_s* fmd_gather;

enum { FMD_GATHER_F32 = 0, FMD_GATHER_PCM16 = 1 };
/* shards on the collector's own device (its own included) also travel through ncclSend / ncclRecv to the self peer instead of a
 * device-to-device copy: exercises the RCCL path where only one GPU is present */
#define FMD_GATHER_LOOPBACK_RCCL 1u
/* The collector ROTATES: block k is gathered on rank (root + k) mod n_ranks.  At throughput-mode rates a single collector would need
 * more than its seven xGMI links carry (83 GB/s asked of each ~77 GB/s link at N = 8) and has ONE PCIe link to hand
 * the result to the host; rotating puts 1 / n_ranks of every rank's output on each link and spreads the hand-over over every GPU's PCIe
 * link.  fmd_gather_wait then returns views on the device of that block's collector (fmd_gather_collector).  Needs one rank per device. */
#define FMD_GATHER_ROTATE 2u

typedef struct {
    int        n_ranks;
    const int* devices;   /* [n_ranks] HIP device ordinal of every rank's handle; ordinals may repeat */
    int        root;      /* collecting rank */
    int        format;    /* FMD_GATHER_F32 | FMD_GATHER_PCM16 */
    unsigned   flags;     /* FMD_GATHER_* */
} fmd_gather_config;

/* handles: [n_ranks] demodulators with the SAME n_channels, block_size and fs_baseband (equal shards: pad the last one with idle
 * stations), handles[r] living on devices[r].  Collective over nothing: called once, by any thread, before the rank threads start. */
int fmd_gather_create(const fmd_gather_config* cfg, const fmd_handle* handles, fmd_gather* out);
/* after every rank thread has left the library (see fmd_gather_abort) */
int fmd_gather_destroy(fmd_gather g);

/* Rank `rank`'s thread, after it has submitted a block to handles[rank] (fmd_submit_*_dev / fmd_process_*_dev): queue the rank's part
 * of the gather of that block — staging copies behind the block's outputs (device-side wait; the library's buffers are released
 * behind them with fmd_release_outputs), then the send (or, on the collecting rank, the receives from every other GPU).
 * Never waits for the GPU; waits on the host only while the rank is two blocks ahead of what the collector has taken with
 * fmd_gather_wait (the collector's three buffer sets). */
int fmd_gather_submit(fmd_gather g, int rank);

/* Collecting rank's thread: block until the oldest block not yet taken has arrived from every rank; DEVICE views (on the
 * collector's device), valid until the next fmd_gather_wait:
 *   audio  [n_ranks * C_local][n_audio][2]  float or int16 (cfg.format)
 *   bytes  [n_ranks * C_local][*cap]        counts [n_ranks * C_local]   (as fmd_rds_bytes_dev) */
int fmd_gather_wait(fmd_gather g, const void** d_audio, const uint8_t** d_rds_bytes, const int** d_rds_counts, int* rds_cap);

/* which rank collects block `block` (0 = the first since fmd_gather_create) and on which device fmd_gather_wait's views of it live */
int fmd_gather_collector(fmd_gather g, long block, int* rank, int* device);
/* Give up, HOST SIDE ONLY: every fmd_gather_submit / fmd_gather_wait that is waiting on the host (for a rank that will never submit, for
 * views that will never be given back) returns FMD_ERR_STATE, now and from here on.  It does not touch the communicators: a rank thread
 * may be inside ncclSend / ncclRecv right now, and ncclCommAbort frees what that call uses.  The communicators are aborted by
 * fmd_gather_destroy — so that a receive whose sender is gone completes and its stream can be drained — which therefore must run AFTER
 * every rank thread has left the library (join them first; fm-radio_amd/host/multi_gpu_host.hpp does).  Any thread. */
int fmd_gather_abort(fmd_gather g);

/* bytes one block moves into the collector from the other GPUs (for sizing against the 7 x ~153 GB/s of xGMI ingress) */
size_t fmd_gather_remote_bytes_per_block(fmd_gather g);
const char* fmd_gather_last_error(fmd_gather g);

#ifdef __cplusplus
}
#endif
#endif
