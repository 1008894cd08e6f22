/* fmdemod.h — C ABI of the MI355X-native batched broadcast-FM demodulator (libfmdemod.so).
 *
 * Drop-in boundary for the demodulator API of williamyang98/FM-Radio (reference src/fm_demod):
 * one fmd_handle is C independent `Broadcast_FM_Demod` instances (reference
 * src/fm_demod/broadcast_fm_demod.h:91-299) advanced in lock-step on one GPU.  Every entry point
 * names the reference member it replaces.  Plain pointers and sizes only; no C++ / torch types.
 *
 * Threading (same contract as the reference): one caller thread per handle.
 *
 * Output lifetime — ONE rule for every output view and getter (fmd_audio_dev, fmd_rds_dev, fmd_get_*): the outputs of a
 * block stay valid while at most FMD_OUTPUT_LIFETIME_BLOCKS (= 5) further fmd_process_* calls have been made on the handle;
 * the call after that reuses the block's buffers.  (The reference keeps one block: "valid until the next Process",
 * broadcast_fm_demod.h:242-256.)  A consumer that may fall further behind tells the library so with
 * fmd_release_outputs(): the library then orders the reuse behind the consumer's stream.
 *
 * Data layouts (channel-major, time contiguous per channel):
 *   IQ in      cf32 [C][N][2] float  (or u8 [C][N][2], RTL-SDR style, converted as `(float)u8 - 127`,
 *              reference src/app.cpp:56-62)
 *   audio out  f32  [C][N_audio][2]  interleaved L,R   (reference Frame<float>, src/audio/frame.h:6-8)
 *   RDS syms   f32  [C][N_rds]  first counts[c] entries valid (reference GetRDSPredSymbols(), .h:253)
 */
#ifndef FMDEMOD_H
#define FMDEMOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMD_API_VERSION 3
/* further fmd_process_* calls during which a block's outputs stay valid (see the lifetime rule above) */
#define FMD_OUTPUT_LIFETIME_BLOCKS 5

typedef struct fmd_handle_s* fmd_handle;

enum {
    FMD_OK = 0,
    FMD_ERR_ARG = -1,        /* bad argument / unsupported configuration */
    FMD_ERR_SIZE = -2,       /* n_samples != block_size or n_channels mismatch: block dropped, nothing emitted
                                (reference Process(): silent return, broadcast_fm_demod.cpp:311-313) */
    FMD_ERR_DEVICE = -3,     /* HIP runtime error, see fmd_last_error */
    FMD_ERR_NO_DEVICE = -4,  /* no usable MI355X: the library has NO CPU fallback */
    FMD_ERR_NAME = -5,       /* unknown stream name */
    FMD_ERR_STATE = -6       /* an earlier fmd_process_* call failed part-way (or the pilot-PLL hand-over watchdog fired): the
                                per-channel state is no longer the state after a whole number of blocks; call fmd_reset */
};

/* reference Broadcast_FM_Demod_Controls::AudioOut (broadcast_fm_demod.h:80) */
enum { FMD_AUDIO_LPR = 0, FMD_AUDIO_LMR = 1, FMD_AUDIO_STEREO = 2 };

/* reference Broadcast_FM_Demod(int block_size) (broadcast_fm_demod.h:229) + the hard-wired
 * Fs_baseband (broadcast_fm_demod.cpp:68), made a parameter. */
typedef struct {
    int n_channels;    /* independent stations on this GPU */
    int block_size;    /* baseband samples per channel per fmd_process call; multiple of 1024*(fs_baseband/256000) */
    int fs_baseband;   /* 256000 (no first decimator), 1024000 (reference), 2048000 */
    int device;        /* HIP device ordinal, -1 = current */
    unsigned flags;    /* FMD_FLAG_* */
} fmd_config;

#define FMD_FLAG_KEEP_TAPS   1u  /* keep the intermediate streams readable through fmd_get_stream */
#define FMD_FLAG_NO_PIPELINE 2u  /* run every stage on the caller's stream, one after the other (debugging / profiling) */
#define FMD_FLAG_PLL_TIME_PARALLEL 4u  /* force the time-parallel pilot-PLL kernel (default: batches <= 7168 channels, and larger ones while stations are out of pilot lock) */
#define FMD_FLAG_PLL_K8           16u  /* time-parallel kernel with 8 (not 16) lanes per channel whatever the batch size (default: effective batches (stations, x 1.5 at 1.024 and 2.048 MSa/s) > 3584; up to 4096 stations 16 lanes all the same while stations are out of pilot lock) */
#define FMD_FLAG_PLL_STREAM_ORDER  32u  /* consecutive blocks' pilot-PLL launches ordered by the stream (kernel boundary) instead of handing over per wavefront while both run (A/B and debugging; same results) */
#define FMD_FLAG_PLL_LOW_WORK      8u  /* force the low-work pilot-PLL kernel (default: larger batches); same results either way */
/* Tolerance mode.  Default (flag clear): every output is bit-identical to the CPU restatement of the reference (oracle/) — the
 * reference's operation order, libm's atan2f, a pilot PLL advanced sample by sample.  That contract has a price the caller should
 * know: a station WITHOUT a lockable pilot (mono station, empty channel, dead input) cannot be advanced under the locked loop's
 * "the frequency word stays put" speculation; its wavefront speculates on the SEQUENCE of words instead (round 6; the serial
 * 78-operation iteration before, ~4x), at about 1.4x a locked wavefront's time, and one such wavefront sets the kernel's duration:
 * 0.89 -> 1.06-1.20 ms per 4096-station block from 1 % unlocked stations on (small batches, where that
 * kernel's latency is the step, pay ~1.5x): a band scan,
 * where most channels are empty, still wants the tolerance mode, whose cost does not depend on lock at all.
 * With the flag the chain keeps the reference's signal flow and state variables but uses cheaper arithmetic: minimax arctangent,
 * hardware sine/cosine, the FIRs as bf16 x 3 products on the matrix cores (fp32 accumulation, ~1e-6 relative), the pilot peak filter
 * as the complex one-pole low-pass of the down-mixed signal that it is, decimated by 16, the pilot PLL advanced 128 samples at a time
 * from the phase at 8 points of the span (the NCO frequency held over the span, the feedback inside it solved exactly on the host),
 * the optional de-emphasis inside the front-end tile (time constants up to ~79 us), and (round 5) the 38 / 57 kHz mixers BEHIND their
 * decimating FIRs: the carrier part of the NCO phase and the Hilbert FIR folded into one complex band-pass FIR per rail, the loop's
 * slow deviation applied once per OUTPUT at the FIR window's centre (the RDS rails with a first-order term in its slope; the L-R rail
 * without: 4e-6 per Hz of NCO offset, i.e. < 1e-5 for a pilot within +-2 Hz of 19 kHz and < 1e-3 for a loop on its +-100 Hz rail).
 * Parity, as tests/test_gpu_fast.py, test_gpu_long.py and test_gpu_realistic.py assert it and profiles/round5/parity_metrics.json records it:
 *   audio, L-R   every block within 1e-4 RMS of the oracle except right behind a sign decision of the reference's L-R phase tracker that
 *                falls on the other side (the allowance is the measured offset difference).  In lock, 64 stations x 30 s: whole-run RMS
 *                <= 2.7e-5 (audio) / 1.4e-5 (L-R) on every station, ONE station-block of 30 016 above 1e-4 (2.7e-4, behind such a decision:
 *                102 of them, 0.053 per station-second; two builds of the reference itself: 0.008-0.042, profiles/round4/reference_flip_evidence.json).
 *                ACQUISITION: over a station's first 0.77 s the whole-run figures are 1.6e-4 (L-R) / 3.1e-4 (audio) on captures whose first
 *                phase estimates flip (4 on 5 stations) — inside the allowance, not inside 1e-4; without flips 1.1e-5 / 2.3e-5.
 *                REALISTIC CAPTURES (noise-like pre-emphasised programme to 15 kHz, L != R, 32 stations x 10 s per condition, u8): CNR 40 /
 *                25 dB, 50 us, carrier +-30 kHz, 110 kHz deviation, pilot +2 Hz, Rician fading, an adjacent station at -20 dB (1.024 MSa/s):
 *                L-R <= 1.2e-5, audio <= 2.9e-5 whole-run in lock on the worst station, L+R <= 2.3e-6; CNR 15 dB: 5.3e-5 / 1.1e-4 (noise moves
 *                single phase estimates in both evaluations).
 *   u8 captures  in a deep fade (a few LSB of signal) consecutive samples fall in exactly opposite directions and the reference's wrap of
 *                a phase difference of exactly pi turns on the last bit of glibc's atan2f — a click of one full turn either way.  The mode
 *                takes the reference's decision from a table made with the exact atan2f (256 kSa/s captures; behind the first decimator
 *                the samples are no longer integers).  Found by the fading condition: L+R 3.4e-3 before, 5.9e-7 with the table.
 *   RDS bits     identical once the synchroniser is in lock (fmd_get_rds_bytes), every station — also on every realistic condition where the
 *                subcarrier stays above the noise; at CNR 15 dB and through fades single symbols are the noise's in either evaluation: the
 *                same groups decode (>= 96 % of the oracle's count on the worst station) and >= 97 % of the bits agree chunk by chunk.
 *   RDS symbols  (fmd_get_rds_symbols, the reference's OnRDSOut payload) the typical symbol within 1.2e-5 and the typical station within
 *                2.6e-5 RMS of the oracle; 0.14 % of the symbols move by 0.1-0.3 where a zero-crossing / clock-wrap decision of the
 *                synchroniser tips (the sign, i.e. the bit, stays) - the reference's own two builds move 1.4 % of theirs; the symbols that
 *                do not move: p99 3.0e-3 on the worst station.  A consumer that needs the soft values inside 1e-4 on EVERY symbol wants
 *                the exact mode.
 * The cost of a block does not depend on the signals: there is no data-dependent path (a station's first 64 ms after a reset also run
 * round 3's per-sample pilot kernel, for the reference's start-up transient; the u8 tie table is consulted on a rare branch).  The
 * FMD_FLAG_PLL_* selectors are ignored.  fmd_default_config() selects this mode. */
#define FMD_FLAG_FAST_MATH        64u

/* reference Broadcast_FM_Demod_Controls (broadcast_fm_demod.h:64-89); defaults in fmd_default_controls */
typedef struct {
    int   audio_out;               /* FMD_AUDIO_* */
    float audio_stereo_mix_factor;
    int   use_deemphasis;
    int   deemphasis_tus;          /* microseconds */
    int   lpr_cutoff_hz;
    int   lmr_cutoff_hz;
} fmd_controls;

/* reference GetBasebandSampleRate() .. GetAudioSampleRate() (broadcast_fm_demod.h:284-288) + block sizes */
typedef struct {
    int fs_baseband, fs_fm_in, fs_fm_out, fs_rds, fs_audio;
    int n_baseband, n_fm_in, n_fm_out, n_rds, n_audio;
} fmd_rates;

/* Every designed coefficient the chain uses (reference filter objects, broadcast_fm_demod.cpp:127-291,
 * bpsk_synchroniser.cpp:26-48).  Same layout as oracle/fm_oracle.h:fmo_coeffs so tests can hand the
 * library's coefficients to the CPU oracle. */
typedef struct {
    int   fs_baseband, m_fm_in;
    float b_fm_in[64], b_fm_out[64], b_hilbert[65];
    float pilot_b[3], pilot_a[3];
    float pll_lpf_b[2], pll_lpf_a[2];
    float deemph_b[2], deemph_a[2];
    float b_lpr[128], b_lmr[128], b_rds[128];
    float ted_lpf_b[2], ted_lpf_a[2];
    float bpsk_lpf_b[2], bpsk_lpf_a[2];
    float fm_gain;
} fmd_coeffs;

int         fmd_api_version(void);
/* == FMD_OUTPUT_LIFETIME_BLOCKS of the library that was loaded */
int         fmd_output_lifetime_blocks(void);
const char* fmd_status_string(int status);
/* number of usable gfx950 devices; <= 0 means every other call fails with FMD_ERR_NO_DEVICE */
int         fmd_device_count(void);

void fmd_default_controls(fmd_controls* c);
/* A configuration for `n_channels` stations of `fs_baseband` with 64 ms blocks (the reference's 65536 samples at 1.024 MSa/s,
 * broadcast_fm_demod.cpp:62-77, scaled to the rate), on the current device, in the mode a many-station deployment wants:
 * FMD_FLAG_FAST_MATH (the tolerance mode above).  `flags = 0`, the bit-exact mode, is what a parity harness asks for explicitly —
 * it costs ~3x per block, and ~13x as soon as 1 % of the stations have no lockable pilot (a band scan: most channels are empty).
 * Returns FMD_ERR_ARG for an unsupported rate. */
int  fmd_default_config(fmd_config* cfg, int n_channels, int fs_baseband);

/* Broadcast_FM_Demod::Broadcast_FM_Demod (broadcast_fm_demod.cpp:59-305) x n_channels */
int fmd_create(const fmd_config* cfg, fmd_handle* out);
/* ~Broadcast_FM_Demod */
int fmd_destroy(fmd_handle h);
/* back to the freshly constructed state (zero histories, AGC gain 0.1, PLLs at rest) */
int fmd_reset(fmd_handle h);

/* GetControls() (broadcast_fm_demod.h:294): channel >= 0 addresses one station, -1 all of them.
 * Takes effect at the next block boundary, like UpdateFilters() (broadcast_fm_demod.cpp:330-389). */
int fmd_set_controls(fmd_handle h, int channel, const fmd_controls* c);
int fmd_get_controls(fmd_handle h, int channel, fmd_controls* c);
int fmd_get_rates(fmd_handle h, fmd_rates* r);
/* the configuration the handle was created with (device resolved to the ordinal in use) */
int fmd_get_config(fmd_handle h, fmd_config* cfg);
int fmd_get_coeffs(fmd_handle h, int channel, fmd_coeffs* k);

/* Broadcast_FM_Demod::Process (broadcast_fm_demod.cpp:309-328) for all channels.
 * *_dev: `d_iq` is a DEVICE pointer.  The call returns without synchronising.  The block is read after everything
 * already queued on `stream` (hipStream_t, NULL = default stream), and work queued on `stream` AFTER the call is
 * ordered behind the library's last read of `d_iq`, so the buffer may be refilled in stream order.  The stages of
 * the block run on the library's own streams and overlap with the neighbouring blocks' stages (up to six blocks in
 * flight); outputs become readable after fmd_synchronize / fmd_wait_outputs and stay valid as the lifetime rule at the
 * top of this header says.  *_host: `iq` is a host pointer; copies, runs, synchronises.
 * `d_iq` may start at any sample (8 / 2 bytes) of an allocation, as the reference's span may; blocks that start on a 16-byte
 * boundary are read fastest (16 bytes per lane).
 * A call that fails with FMD_ERR_DEVICE after some of its kernels were queued leaves the handle in FMD_ERR_STATE. */
int fmd_process_cf32_dev(fmd_handle h, const float* d_iq, int n_channels, int n_samples, void* stream);
int fmd_process_u8_dev(fmd_handle h, const uint8_t* d_iq, int n_channels, int n_samples, void* stream);
int fmd_process_cf32_host(fmd_handle h, const float* iq, int n_channels, int n_samples);
int fmd_process_u8_host(fmd_handle h, const uint8_t* iq, int n_channels, int n_samples);
/* The same blocks submitted WITHOUT touching the caller's streams (API v3) — for hosts that rotate several input buffers, as the
 * reference's device thread does with its USB buffers (src/device/device.cpp:107-119) and fm-radio_amd/host/station_ring.hpp does
 * with its staging blocks.  The block is read after everything already queued on `ready_stream` (NULL: the data is in place
 * now); nothing is queued on `ready_stream` or any other stream of the caller, so the caller's streams never wait for the
 * demodulator and consecutive blocks' first stages run back to back (fmd_process_*_dev orders the caller's stream behind the
 * library's read of every block, which also orders the next block's submission behind it: two cross-queue hand-overs, ~0.1 ms,
 * between consecutive front-end launches).  The buffer may be rewritten once fmd_wait_input() says so. */
int fmd_submit_cf32_dev(fmd_handle h, const float* d_iq, int n_channels, int n_samples, void* ready_stream);
int fmd_submit_u8_dev(fmd_handle h, const uint8_t* d_iq, int n_channels, int n_samples, void* ready_stream);
/* make `stream` wait (on the device, no host block) until the library has finished reading the input buffer of the newest
 * block submitted with fmd_submit_*_dev / fmd_process_*_dev */
int fmd_wait_input(fmd_handle h, void* stream);
int fmd_synchronize(fmd_handle h);
/* make `stream` wait (on the device, no host block) until the newest block's outputs are complete */
int fmd_wait_outputs(fmd_handle h, void* stream);
/* Which outputs the device-side calls (fmd_wait_outputs, fmd_release_outputs, fmd_audio_dev, fmd_audio_pcm16_dev, fmd_rds_dev,
 * fmd_rds_bytes_dev) refer to behind fmd_submit_*_dev in the tolerance mode.
 * Background: there, with 1024 stations' worth of 256 kSa/s blocks or more, a block's extract and RDS stages are queued when the NEXT
 * block is submitted — behind that block's front end, on the same hardware queue: the two large kernels take turns instead of sharing
 * the CUs (6 % on the step, DESIGN.md "Schedule") — or as soon as somebody needs them.
 *   on = 0 (default): the NEWEST block's outputs.  A device-side call that wants them while they are still put off queues them at
 *       once, and from then on the handle queues every block's stages at submission (a consumer that asks after every block would
 *       otherwise stall the front end's queue each time).
 *   on = 1: the newest QUEUED outputs, never forcing anything: after fmd_submit_*_dev of block k those are block k - 1's where the
 *       stages are put off, block k's otherwise — fmd_outputs_block says which (nothing before the second block at most: fmd_wait_outputs / fmd_release_outputs do nothing then, fmd_audio_pcm16_dev fails with FMD_ERR_ARG).
 *       A consumer that takes every block's outputs one submission later (bench.py's per-step gather does) keeps the faster schedule.
 * fmd_synchronize, the host getters and fmd_process_* always complete the newest block.  Synchronises; call it between blocks. */
int fmd_set_output_lag(fmd_handle h, int on);
/* which block the device-side output calls refer to right now: 0 = the first block since fmd_create / fmd_reset, -1 = none yet.
 * (Under fmd_set_output_lag(h, 1) a consumer needs it to tell k from k - 1: batches below the size named above, and the exact mode,
 * queue every block's stages at submission.) */
int fmd_outputs_block(fmd_handle h, long* block);
/* how often the handle's block numbering has restarted (fmd_create counts as the first start; every fmd_reset restarts it): a consumer
 * that follows fmd_outputs_block across resets — the multi-GPU gather does — tells a restart from a repeated or skipped block by it */
int fmd_outputs_epoch(fmd_handle h, long* epoch);
/* The consumer's side of the lifetime rule: everything queued on `stream` so far (the kernels / copies that read the newest
 * block's output views) must finish before the library overwrites those views, however many blocks are submitted meanwhile.
 * Records an event on `stream`; the library's writers of that buffer slot wait for it on the device.  Never blocks the host. */
int fmd_release_outputs(fmd_handle h, void* stream);

/* OnAudioOut() / GetAudioOut() (broadcast_fm_demod.h:256,297): device views of the current block */
int fmd_audio_dev(fmd_handle h, const float** d_audio /* [C][n_audio][2] */);
/* The newest block's audio as the 16-bit PCM frames the reference's headless scraper writes to its WAV files
 * (fm_scraper.cpp:79-82: sample * (32767 * 0.95f), truncated toward zero): d_pcm [C][n_audio][2] int16 on the device,
 * converted on `stream` once the block's outputs are complete.  Half the bytes of the f32 block: the payload of the
 * multi-GPU audio gather. */
int fmd_audio_pcm16_dev(fmd_handle h, int16_t* d_pcm /* [C][n_audio][2] */, void* stream);
/* OnRDSOut() / GetRDSPredSymbols() (broadcast_fm_demod.h:253,298) */
int fmd_rds_dev(fmd_handle h, const float** d_syms /* [C][n_rds] */, const int** d_counts /* [C] */);
/* host copies (synchronise first) */
int fmd_get_audio(fmd_handle h, float* audio /* [C][n_audio][2] */);
int fmd_get_rds_symbols(fmd_handle h, float* syms /* [C][n_rds] */, int* counts /* [C] */);

/* Parity / GUI taps, the reference's buffer getters (broadcast_fm_demod.h:242-256, :291):
 *   "fm_out_iq" GetFMOutIQ [C][n_fm_out][2] | "pll_dt" [C][n_fm_out] | "lpr" GetLPRAudioOutput [C][n_audio]
 *   "lmr" GetLMRAudioOutput [C][n_audio] | "rds" GetRDSOutput (post-AGC) [C][n_rds][2]
 *   "rds_raw_sym" GetRDSRawSymbols [C][n_rds][2] | "lmr_phase" GetAudioLMRPhaseError [C]
 *   "agc_pilot_gain" [C] | "agc_rds_gain" [C]
 *   exact mode with FMD_FLAG_KEEP_TAPS, the two loops' per-sample traces (broadcast_fm_demod.h:245-248, bpsk_synchroniser.h:78-85):
 *   "pilot" GetPilotOutput (after its AGC) [C][n_fm_out][2] | "pll" GetPLLOutput (cos, sin) [C][n_fm_out][2]
 *   "pll_raw_err" Get_PLL_Raw_Phase_Error_Output [C][n_fm_out] | "pll_pi_err" Get_PLL_LPF_Phase_Error_Output [C][n_fm_out]
 *   "bpsk_pll_sym" GetPLLSymbols [C][n_rds][2] | "bpsk_intdump" GetIntDumpFilter [C][n_rds][2] | "bpsk_ted_raw" GetTEDRawPhaseError,
 *   "bpsk_ted_pi" GetTEDPIPhaseError, "bpsk_pll_raw" GetPLLRawPhaseError, "bpsk_pll_pi" GetPLLPIPhaseError, "bpsk_zcd" GetZeroCrossings and
 *   "bpsk_trig" GetIntDumpTriggers (the two bool traces as 0 / 1) [C][n_rds] — bit-identical
 *   to the oracle's; the tolerance mode evaluates its loops at eight points per span / on groups of four samples and has no such trace
 *   (FMD_ERR_NAME)
 *   FMD_FLAG_FAST_MATH keeps fm_out alone (the kernels that need the Hilbert rail make it for themselves) and the NCO phase as one
 *   cubic per 128 samples: "fm_out_iq" and "pll_dt" then need FMD_FLAG_KEEP_TAPS too; "pll_poly" [C][1 + n_fm_out / 128][4] (tolerance mode only) is always there
 * Copies the current block's values to `out` (host); *n_floats receives the float count.  Needs
 * FMD_FLAG_KEEP_TAPS for the streams a fused pipeline would not otherwise materialise. */
int fmd_get_stream(fmd_handle h, const char* name, float* out, size_t cap_floats, size_t* n_floats);

const char* fmd_last_error(fmd_handle h);

/* Per-channel state snapshot / restore (SURVEY.md §5 "checkpoint / resume"; what moving a station between handles or GPUs
 * needs): every filter history, AGC gain, loop integrator, NCO phase, BPSK synchroniser and Manchester decoder variable of ONE
 * channel — the member variables of one reference Broadcast_FM_Demod (broadcast_fm_demod.h:94-227) — as an opaque,
 * self-describing blob of fmd_state_size() bytes.  Both calls synchronise the handle first.  A blob can be restored into any
 * channel of any handle with the same fs_baseband; the restored channel then continues bit-identically.  Controls are not
 * part of the blob (fmd_get_controls / fmd_set_controls). */
size_t fmd_state_size(fmd_handle h);
int    fmd_get_state(fmd_handle h, int channel, void* blob, size_t cap_bytes);
int    fmd_set_state(fmd_handle h, int channel, const void* blob, size_t n_bytes);

/* Differential Manchester decode of the RDS symbol stream on the GPU, one decoder per channel
 * (reference src/rds_decoder/differential_manchester_decoder.h:25-60; the 16-byte buffer size is the
 * one src/app.cpp:13-20 uses).  bytes: [C][cap_bytes]; counts[c] = bytes appended for channel c this
 * block (multiples of 16). */
int fmd_get_rds_bytes(fmd_handle h, uint8_t* bytes, int cap_bytes_per_channel, int* counts);
/* ... and as device views of the newest block (same lifetime rule as fmd_audio_dev): d_bytes [C][*cap_bytes_per_channel],
 * d_counts [C] — for hosts that fetch the outputs with their own asynchronous copies (fm-radio_amd/host/station_ring.hpp). */
int fmd_rds_bytes_dev(fmd_handle h, const uint8_t** d_bytes, const int** d_counts, int* cap_bytes_per_channel);

/* ------------------------------------------------------------------------------------------------------------------
 * Wideband channeliser (SURVEY.md §8f row 3 / BASELINE configs[4]; NOT part of the reference, which tunes one station in
 * the RTL-SDR hardware): splits one wideband cf32 capture into n_stations channels at fs_out, laid out [C][n_out] cf32 —
 * the input layout of fmd_process_cf32_dev.  Per station: mix the centre frequency to 0, then a rational polyphase
 * decimator L/M = fs_out/fs_in (256 k / 10 M = 16 / 625) built from one Kaiser-windowed prototype (cut-off fs_out / 2,
 * 60 dB).  Streaming: histories and the mixers' phases carry over from call to call.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct fmd_channelizer_s* fmd_channelizer;
typedef struct {
    double        fs_in;             /* wideband sample rate, integer Hz */
    double        fs_out;            /* per-station rate handed to the demodulator, integer Hz (256000) */
    int           n_stations;
    const double* center_hz;         /* [n_stations] station centres relative to the capture's centre, |f| < fs_in / 2 */
    int           taps_per_phase;    /* 0 = default (640), multiple of 4 */
    long long     max_input_samples; /* largest n_in of a process call */
    int           device;            /* HIP device ordinal, -1 = current */
} fmd_chan_config;

/* host-only filter design (no GPU needed): L/M and, if taps != NULL, the prototype stored [t][p] (t = tap within phase p) */
int fmd_chan_design(double fs_in, double fs_out, int taps_per_phase, float* taps, int* L, int* M);
int fmd_chan_create(const fmd_chan_config* cfg, fmd_channelizer* out);
int fmd_chan_destroy(fmd_channelizer h);
int fmd_chan_reset(fmd_channelizer h);
int fmd_chan_info(fmd_channelizer h, int* L, int* M, int* taps_per_phase, int* n_stations);
int fmd_chan_get_taps(fmd_channelizer h, float* taps, size_t cap_floats);
/* d_wide: [n_in][2] cf32 on the device; n_in * L must be a multiple of M (625 input samples per 16 outputs at 10 M -> 256 k).
 * d_out: [n_stations][out_capacity_per_station][2] cf32 on the device — station k's *n_out samples start at row k (row
 * stride = out_capacity_per_station; pass the exact n_out to get the dense [C][n_out] layout fmd_process_cf32_dev takes).
 * Asynchronous on `stream`; consecutive calls may use different streams (the library orders them).  d_wide is read IN PLACE by the call's
 * kernel (no staging copy): like d_out it belongs to the call until its work on `stream` has completed. */
int fmd_chan_process_cf32_dev(fmd_channelizer h, const float* d_wide, size_t n_in, float* d_out, size_t out_capacity_per_station,
                              size_t* n_out, void* stream);
const char* fmd_chan_last_error(fmd_channelizer h);

#ifdef __cplusplus
}
#endif
#endif
