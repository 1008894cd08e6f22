// C ABI of libfmdemod.so (include/fmdemod.h): handle management, coefficient design, per-channel
// controls, device buffers and the per-block kernel launch sequence.  Host side only; the kernels are
// in fmd_kernels.hip.  There is no CPU fallback: without a gfx950 device every call fails loudly.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <complex>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "fmd_design.h"
#include "fmd_kernels.h"
#include "fmd_math.h"
#include "fmdemod.h"
#include "fmdemod_debug.h"

using namespace fmd;

// Stage placement: k_front on sF, k_pilot_power on sA, k_pilot_pll on sB, k_extract (+ k_lmr_phase) on sX, k_rds_sync on sR.
// Blocks rotate through kSlots buffer slots and every stage waits only for its producer (HIP events), so in steady state
// all stages run concurrently on different blocks.
enum Stage { ST_FRONT = 0, ST_DEEMPH, ST_POWER, ST_PLL, ST_EXTRACT, ST_RDS, ST_PREDECIM, ST_COUNT };
static const char* const kStageName[ST_COUNT] = {"k_front", "k_deemphasis+k_hilbert", "k_pilot_power", "k_pilot_pll", "k_extract", "k_rds_sync", "k_predecim"};
// ... and of the tolerance mode's kernels, as they appear in rocprofv3 kernel traces
static const char* const kStageNameFast[ST_COUNT] = {"k_front_mfma", "k_deemphasis", "k_pilot_power", "k_pll_sparse", "k_extract_bp", "k_rds_sync", "k_predecim_mfma"};

struct ProfiledBlock { hipEvent_t t0[ST_COUNT], t1[ST_COUNT]; bool used[ST_COUNT]; bool chain = false; };

struct fmd_handle_s {
    fmd_config cfg{};
    LaunchCtx ctx{};
    fmd_coeffs base{};                       // coefficients common to all channels (+ channel-0 control designs)
    std::vector<fmd_controls> controls;      // per channel
    bool controls_dirty = true;
    bool deemph_on = false;                  // some channel uses the de-emphasis IIR
    bool deemph_linger = false;              // keep that path one more block after the last channel leaves it, so the
                                             // Hilbert FIR sees the de-emphasised history the reference would hold
    std::vector<void*> allocs;
    void* d_in = nullptr;                    // staging for the host-pointer entry points
    size_t d_in_bytes = 0;
    hipStream_t own_stream = nullptr;        // host-pointer entry points, uploads, resets
    hipStream_t last_stream = nullptr;
    hipStream_t sF = nullptr, sA = nullptr, sB = nullptr, sB2 = nullptr, sX = nullptr, sR = nullptr, sD = nullptr;   // sD: the optional de-emphasis stage
    unsigned pll_seq = 0;                    // k_pilot_pll launches handed over per wavefront so far (0: hand-over by stream order)
    // exact mode, 3585 .. 4096 effective stations: the pilot-PLL kernel's lane count follows what is out of lock — 8 lanes a station while every
    // loop holds lock (fewest instructions beside the FIR kernels), 16 while some do not (a loop out of lock costs its wavefront ~1.8x, and
    // the kernel lasts as long as its slowest wavefront: 1.55 -> 1.12 ms a block with 16).  The kernel counts the wavefronts that ran out
    // of lock (Buffers::pll_hint[C]); a 4-byte copy of the counter comes back every other block, and the host looks at it without waiting.
    // The host may be queueing many blocks ahead of the GPU: what it goes by is the pair that came back — the newest launch that had wavefronts out
    // of lock and the newest launch that has run: out of lock "now" while the two are fewer than 8 launches apart; no news, no change.
    bool pll_k_adaptive = false;
    unsigned* pll_unl_host = nullptr;        // pinned: [0] newest launch with wavefronts out of lock, [1] newest launch that has run
    bool pll_chained = false;
    int pll_waves = 0;
    hipEvent_t ev_in = nullptr, ev_P[kSlots] = {}, ev_F[kSlots] = {}, ev_A[kSlots] = {}, ev_B[kSlots] = {}, ev_E[kSlots] = {}, ev_X[kSlots] = {};
    hipEvent_t ev_D[kSlots] = {};            // k_front done, de-emphasis stage may start
    hipEvent_t ev_C[kSlots] = {};            // fmd_release_outputs: the consumer of a slot's outputs has finished with them
    bool consumer_pending[kSlots] = {};
    bool slot_used[kSlots] = {};
    bool last_block_deemph = false;          // the previous block went through the de-emphasis stage (stream sD)
    bool poisoned = false;                   // a block failed part-way: state is not the state after a whole number of blocks
    bool pipelined = true;
    // Tolerance mode, fmd_submit_*: k_extract_bp shares k_front_mfma's stream and a block's extract + RDS stages are queued when the
    // NEXT block is submitted (behind that block's front end) or when somebody asks for the outputs — see process_dev
    bool lazy_extract = false, lazy_capable = false;
    bool no_fused_pll = false;               // development A/B: the deferred pilot stage as a launch of its own
    bool uniform_cutoffs = true;             // every station has the same L+R / L-R cut-offs (one set of tap tables): what k_chain needs of the controls
    bool chain_off = true;                   // the one-launch form of a steady block (k_chain) is OFF unless fmd_debug_set_chain(h, 1) / FMD_CHAIN=1 (development builds) asks for it: measured slower, DESIGN.md section 3
    long chain_blocks = 0;                   // blocks run as k_chain since create / reset
    bool last_block_chain = false;
    bool pll_eager = false;                  // development A/B: the pilot stage queued at submission on its own queue (round 3's arrangement)
    bool split_queues = false;               // the front end's and the extract stage's queues on disjoint sets of CUs (hipExtStreamCreateWithCUMask): the two run side by side
    bool front_with_predecim = true;         // 1.024 / 2.048 MSa/s, deferred schedule: the front end follows the first decimator on ITS queue
    // pll_pending: the block's pilot stage has not been queued either — it rides in the next block's front-end launch (k_front_mfma<FUSED>) or,
    // where that is not possible (a start-up block, the getters' per-sample streams, a flush), goes in front of the extract stage on its own;
    // front_dep: the event behind the stage that made the block's fm_out; pll_dep / pll_stream: where the pilot stage was queued (NULL event: same
    // queue as the front end, nothing to wait for)
    struct Deferred { bool active = false, pll_pending = false, front_cross = false, deemph = false; SlotRef ref{}; hipEvent_t pll_dep = nullptr, front_dep = nullptr; hipStream_t pll_stream = nullptr, front_stream = nullptr;
                      int slot = 0; long block = 0; ProfiledBlock* pm = nullptr; bool prof_p = false, prof_x = false, prof_r = false; } deferred;
    hipStream_t last_x_stream = nullptr;     // where the newest extract stage was queued, and the event behind it: consecutive blocks'
    hipEvent_t last_x_event = nullptr;       // extract stages are ordered (L-R phase estimate), whichever of the two streams they take
    hipStream_t last_p_stream = nullptr;     // where the newest pilot stage was queued and the event behind it: consecutive blocks' pilot stages run in
    hipEvent_t last_p_event = nullptr;       // order (loop state), whichever queue each takes (its own, or the front end's); NULL: drained
    hipEvent_t x_done[kSlots] = {};          // fires when the extract stage of the block in that slot has run (its timing event while it is being timed); NULL: drained
    int warm_left = 0;                       // tolerance mode: blocks still inside some station's start-up transient (k_pll_span runs beside k_pll_sparse)
    long n_blocks = 0;                       // blocks submitted since create/reset; slot = n_blocks % kSlots
    int out_slot = 0;                        // slot holding the newest outputs (the newest block's; under fmd_set_output_lag: the newest QUEUED outputs)
    int sub_slot = 0;                        // slot of the newest submitted block
    bool have_out = false;                   // some block's output stages have been queued since create / reset
    long out_block = -1;                     // ... and which block's (0 = the first since create / reset) the output views are
    long epoch = 0;                          // how often the block numbering restarted (fmd_reset, fmd_set_config ...): fmd_outputs_epoch
    bool lag_outputs = false;                // fmd_set_output_lag
    hipEvent_t ev_consumed = nullptr;        // fires when the newest block's input buffer has been read (fmd_wait_input)
    int device = 0;
    int bytes_cap = 0;
    std::string err;
    std::map<int, std::vector<float>> lpf_cache;  // cut-off Hz -> 128 taps
    std::map<int, int> img_slot;                  // FMD_FLAG_FAST_MATH: cut-off Hz -> slot of its operand image in aud_img (k_extract_bp)
    size_t img_capacity = 0;                      // slots allocated in aud_img
    unsigned debug_skip = 0;                 // development knob FMD_DEBUG_SKIP_STAGES: bit (1 << Stage) = do not launch that stage (timing experiments only: outputs are garbage)
    int profiling = 0;                       // 0 off, 1 every kernel of every block, 2 k_pilot_pll every block + the rest every 4th
    std::vector<ProfiledBlock*> marks;       // one per profiled block, drained by fmd_profile_read
};

namespace {

// The pipeline keeps six streams busy beside the caller's; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default
// 4) and dependent stages sharing a queue block each other.  The library does not touch the process environment: the host
// application sets GPU_MAX_HW_QUEUES >= 8 before the HIP runtime initialises, and fmd_create warns once on
// stderr (and in fmd_last_error(NULL)) when it finds less.  FMD_QUIET=1 silences the warning.
void warn_hw_queues_once() {
    static bool done = false;
    if (done) return;
    done = true;
    const char* v = getenv("GPU_MAX_HW_QUEUES");
    const int n = v ? atoi(v) : 4;
    if (n >= 7) return;
    const char* q = getenv("FMD_QUIET");
    if (q && *q && *q != '0') return;
    fprintf(stderr, "libfmdemod: GPU_MAX_HW_QUEUES=%s: the pipelined demodulator uses up to 7 streams and loses its overlap on fewer hardware "
                    "queues; set GPU_MAX_HW_QUEUES=16 in the environment before the HIP runtime initialises (8 if nothing else in the process "
                    "creates streams; see INTEGRATION.md)\n", v ? v : "(unset, default 4)");
}

thread_local std::string g_create_error;

int fail(fmd_handle h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail((h), FMD_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int dev_alloc(fmd_handle h, T** p, size_t count) {
    void* q = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    HIP_TRY(h, hipMalloc(&q, bytes));
    // on the handle's own stream: a null-stream hipMemset is not ordered against the (non-blocking) own_stream, and
    // could land after the k_reset / control uploads that follow
    HIP_TRY(h, hipMemsetAsync(q, 0, bytes, h->own_stream));
    h->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return FMD_OK;
}

constexpr unsigned kKnownFlags = FMD_FLAG_KEEP_TAPS | FMD_FLAG_NO_PIPELINE | FMD_FLAG_PLL_TIME_PARALLEL | FMD_FLAG_PLL_LOW_WORK | FMD_FLAG_PLL_K8 |
                                 FMD_FLAG_PLL_STREAM_ORDER | FMD_FLAG_FAST_MATH;

bool config_ok(const fmd_config* c, int* m) {
    if (!c || c->n_channels <= 0) return false;
    if (c->flags & ~kKnownFlags) return false;        // a flag this build does not implement must not be silently ignored
    if (c->fs_baseband != 256000 && c->fs_baseband != 1024000 && c->fs_baseband != 2048000) return false;
    *m = c->fs_baseband / 256000;
    if (c->block_size <= 0 || (c->block_size % (1024 * *m)) != 0) return false;
    return true;
}

// Operand images of k_front_mfma's FIRs (fmd_kernels.hip FrontGeomM): v_mfma_f32_16x16x32_bf16's A operand, lane l = row l % 16,
// k = 8 (l / 16) + 0..7; A[m][t] = taps[t - stride m] inside the band, 0 outside; every fp32 tap as two bf16 (round to nearest even).
static uint16_t bf16_rne(float x) {
    uint32_t u; std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float x; std::memcpy(&x, &u, 4); return x; }
// one FIR's image: A[m][t] = taps[t - shift - stride m], t < 32 ksteps; [k-step][hi / lo][lane][8]
void toeplitz_image(const float* taps, int n_taps, int stride, int ksteps, uint16_t* img, int shift = 0) {
    for (int sK = 0; sK < ksteps; sK++)
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 8; i++) {
                const int t = 32 * sK + 8 * (l / 16) + i, idx = t - shift - stride * (l % 16);
                const float v = (idx >= 0 && idx < n_taps) ? taps[idx] : 0.0f;
                const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f32(hi));
                img[(((size_t)sK * 2 + 0) * 64 + l) * 8 + i] = hi;
                img[(((size_t)sK * 2 + 1) * 64 + l) * 8 + i] = lo;
            }
}

// k_extract_bp (fmd_kernels_bp.inc): the harmonic mixer folded into the decimating FIR.  The reference's L-R / RDS rails are
//   sum_tau h[tau] a[t0 + tau] e^{j 2 pi H dt[t0 + tau]},  a[t] = x[t - 32] + j sum_n b_hil[n] x[t - 64 + n]   (broadcast_fm_demod.cpp:463-536)
// and the NCO's carrier part e^{-j 2 pi H 19 (t + 1) / 128} with the Hilbert FIR is ONE complex FIR of 192 taps on the real signal x:
//   G[u] = sum_tau h[tau] e^{-j 2 pi H 19 tau / 128} (delta[u - tau - 32] + j b_hil[u - tau]),     S[m] = sum_u G[u] x[M m - 188 + 4 (M / 8) + u]
// (in double; the phase of the carrier through its exact period of 128 samples)
constexpr int kBpTaps = 192;
void bandpass_taps(const float* h, const float* b_hil, int H, float* g_re, float* g_im) {
    double re[kBpTaps] = {0.0}, im[kBpTaps] = {0.0};
    const double two_pi = 6.283185307179586476925;
    for (int tau = 0; tau < 128; tau++) {
        const int k = (H * 19 * tau) % 128;
        const double cr = std::cos(two_pi * k / 128.0), ci = -std::sin(two_pi * k / 128.0), hv = h[tau];
        re[tau + 32] += hv * cr; im[tau + 32] += hv * ci;
        for (int n = 0; n < 65; n++) {
            const double b = b_hil[n];
            if (b == 0.0) continue;
            re[tau + n] += hv * (-ci) * b;        // j (cr + j ci) = -ci + j cr
            im[tau + n] += hv * cr * b;
        }
    }
    for (int u = 0; u < kBpTaps; u++) { g_re[u] = (float)re[u]; g_im[u] = (float)im[u]; }
}
// Tap tables of k_extract_bp (fmd_kernels_bp.inc): kBpTabTL bf16 each, tap i at element kBpTabPadL + i (+ 4 for a "copy 4" table)
constexpr int kBpTabPadL = 96, kBpTabTL = 448;
constexpr size_t kBpTabSlotU16 = (size_t)12 * kBpTabTL, kBpRdsTabU16 = (size_t)6 * kBpTabTL;
void bp_tap_table(const float* taps, int n_taps, int move, uint16_t* hi, uint16_t* lo) {
    for (int i = 0; i < kBpTabTL; i++) { hi[i] = 0; if (lo) lo[i] = 0; }
    for (int i = 0; i < n_taps; i++) {
        const uint16_t h = bf16_rne(taps[i]);
        hi[kBpTabPadL + move + i] = h;
        if (lo) lo[kBpTabPadL + move + i] = bf16_rne(taps[i] - bf16_to_f32(h));
    }
}
// a stride-4 family's block: [hi, copy 0][hi, copy 4][lo, copy 0][lo, copy 4]
void bp_family_block(const float* taps, int n_taps, uint16_t* dst) {
    bp_tap_table(taps, n_taps, 0, dst, dst + 2 * kBpTabTL);
    bp_tap_table(taps, n_taps, 4, dst + kBpTabTL, dst + 3 * kBpTabTL);
}
void bp_slot_tap_tables(const float* taps, const float* b_hil, uint16_t* dst) {
    bp_family_block(taps, 128, dst);
    float gr[kBpTaps], gi[kBpTaps];
    bandpass_taps(taps, b_hil, 2, gr, gi);
    bp_family_block(gr, kBpTaps, dst + 4 * kBpTabTL);
    bp_family_block(gi, kBpTaps, dst + 8 * kBpTabTL);
}
// RDS (rows (o, rail), band offset 4 + 8 o: every table moved on by four): S0 [re hi][im hi][re lo][im lo], S1 [re hi][im hi]
void bp_rds_tap_tables(const float* b_rds, const float* b_hil, uint16_t* dst) {
    float gr[kBpTaps], gi[kBpTaps], h1[128];
    bandpass_taps(b_rds, b_hil, 3, gr, gi);
    bp_tap_table(gr, kBpTaps, 4, dst, dst + 2 * kBpTabTL);
    bp_tap_table(gi, kBpTaps, 4, dst + kBpTabTL, dst + 3 * kBpTabTL);
    for (int i = 0; i < 128; i++) h1[i] = (float)(((double)i - 63.5) * (double)b_rds[i]);
    bandpass_taps(h1, b_hil, 3, gr, gi);
    bp_tap_table(gr, kBpTaps, 4, dst + 4 * kBpTabTL, nullptr);
    bp_tap_table(gi, kBpTaps, 4, dst + 5 * kBpTabTL, nullptr);
}

// The block's first 31 L-R outputs, the part of their sums that lies in the PREVIOUS block (mixed with its L-R offset, reference
// broadcast_fm_demod.cpp:485-517): S_old[m] = sum_{tau < 124 - 4 m} h[tau] e^{-j 2 pi 38 tau / 128} a[4 m - 124 + tau] = sum_u K_m[u] W[u],
// W[u] = fm_out[u - 188], K_m[u] = (composite of the TRUNCATED taps)[u - 4 m].  Layout [m][lane p of 8][24 columns][re, im], fp16.
constexpr size_t kBpEdgeHalves = (size_t)31 * 8 * 24 * 2;
static uint16_t f32_to_f16_rne(float x) {       // IEEE binary16, round to nearest even (the matrix entries are < 1: no overflow; denormals kept)
    uint32_t u; std::memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int e = (int)((u >> 23) & 0xffu) - 127 + 15;
    uint32_t m = u & 0x7fffffu;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int sh = 14 - e;
        const uint32_t r = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        return (uint16_t)(sign | (r + ((rem > half || (rem == half && (r & 1u))) ? 1u : 0u)));
    }
    const uint32_t r = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1fffu;
    return (uint16_t)(sign | (r + ((rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ? 1u : 0u)));
}
void bp_edge_matrix(const float* taps, const float* b_hil, uint16_t* dst) {
    std::memset(dst, 0, sizeof(uint16_t) * kBpEdgeHalves);
    for (int m = 0; m < 31; m++) {
        float ht[128], gr[kBpTaps], gi[kBpTaps];
        for (int i = 0; i < 128; i++) ht[i] = i < 124 - 4 * m ? taps[i] : 0.0f;
        bandpass_taps(ht, b_hil, 2, gr, gi);
        for (int u = 0; u < 192; u++) {
            const int idx = u - 4 * m;
            if (idx < 0 || idx >= kBpTaps) continue;
            uint16_t* e = dst + (((size_t)m * 8 + u / 24) * 24 + u % 24) * 2;
            e[0] = f32_to_f16_rne(gr[idx]); e[1] = f32_to_f16_rne(gi[idx]);
        }
    }
}

const std::vector<float>& lpf_taps(fmd_handle h, int hz) {
    auto it = h->lpf_cache.find(hz);
    if (it != h->lpf_cache.end()) return it->second;
    fmd_controls c;
    fmd_default_controls(&c);
    c.lpr_cutoff_hz = hz;
    fmd_coeffs k{};
    design_controls(&k, &c);
    return h->lpf_cache.emplace(hz, std::vector<float>(k.b_lpr, k.b_lpr + 128)).first->second;
}

// reference UpdateFilters() (broadcast_fm_demod.cpp:330-389), for every channel whose controls changed
int upload_controls(fmd_handle h, hipStream_t s) {
    const int C = h->cfg.n_channels;
    std::vector<float> lpr((size_t)C * 128), lmr((size_t)C * 128), de((size_t)C * 4), mix((size_t)C * 2);
    int any = 0;
    bool slow_pole = false;   // a de-emphasis time constant beyond what k_front's in-tile form covers (fmd_kernels.hip kDeemphWarmup)
    for (int c = 0; c < C; c++) {
        const fmd_controls& k = h->controls[c];
        const auto& a = lpf_taps(h, k.lpr_cutoff_hz);
        const auto& b = lpf_taps(h, k.lmr_cutoff_hz);
        std::copy(a.begin(), a.end(), lpr.begin() + (size_t)c * 128);
        std::copy(b.begin(), b.end(), lmr.begin() + (size_t)c * 128);
        fmd_coeffs kk{};
        design_controls(&kk, &k);  // cheap: two cached-size designs would be overkill to cache for the 1-pole
        de[4 * c + 0] = kk.deemph_b[0]; de[4 * c + 1] = kk.deemph_b[1]; de[4 * c + 2] = kk.deemph_a[0];
        de[4 * c + 3] = k.use_deemphasis ? 1.0f : 0.0f;
        any |= k.use_deemphasis ? 1 : 0;
        if (k.use_deemphasis && !(kk.deemph_a[0] >= 0.0f && kk.deemph_a[0] <= 0.905f)) slow_pole = true;
        mix[2 * c] = (float)k.audio_out; mix[2 * c + 1] = k.audio_stereo_mix_factor;
    }
    Buffers& b = h->ctx.b;
    if (h->ctx.fast) {
        // k_extract_bp: one set of tap tables per distinct audio cut-off, and which two each station uses
        std::vector<int> idx((size_t)C * 2);
        bool grew = false;
        for (int c = 0; c < C; c++)
            for (int which = 0; which < 2; which++) {
                const int hz = which ? h->controls[c].lmr_cutoff_hz : h->controls[c].lpr_cutoff_hz;
                auto it = h->img_slot.find(hz);
                if (it == h->img_slot.end()) { it = h->img_slot.emplace(hz, (int)h->img_slot.size()).first; grew = true; }
                idx[(size_t)c * 2 + which] = it->second;
            }
        if (grew) {
            const size_t n_slots = h->img_slot.size();
            if (n_slots > h->img_capacity) {   // (everything is idle here: upload_controls runs behind sync_all)
                const size_t cap = std::max<size_t>(2 * n_slots, 8);
                uint4* q4 = nullptr;
                int rc = dev_alloc(h, &q4, cap * kBpTabSlotU16 * 2 / sizeof(uint4));
                if (rc) return rc;
                b.bp_tab = q4;       // the old table stays on the handle's allocation list until fmd_destroy
                uint4* q3 = nullptr;
                rc = dev_alloc(h, &q3, cap * kBpEdgeHalves * 2 / sizeof(uint4));
                if (rc) return rc;
                b.bp_edge = q3;
                h->img_capacity = cap;
            }
            // per cut-off the tap tables of the L+R FIR and of the L-R composite (mixer + Hilbert FIR folded into the taps)
            std::vector<uint16_t> tabsv(n_slots * kBpTabSlotU16);
            for (const auto& kv : h->img_slot) bp_slot_tap_tables(lpf_taps(h, kv.first).data(), h->base.b_hilbert, tabsv.data() + (size_t)kv.second * kBpTabSlotU16);
            HIP_TRY(h, hipMemcpyAsync(b.bp_tab, tabsv.data(), tabsv.size() * 2, hipMemcpyHostToDevice, s));
            std::vector<uint16_t> edges(n_slots * kBpEdgeHalves);
            for (const auto& kv : h->img_slot) bp_edge_matrix(lpf_taps(h, kv.first).data(), h->base.b_hilbert, edges.data() + (size_t)kv.second * kBpEdgeHalves);
            HIP_TRY(h, hipMemcpyAsync(b.bp_edge, edges.data(), edges.size() * 2, hipMemcpyHostToDevice, s));
            HIP_TRY(h, hipStreamSynchronize(s));
        }
        HIP_TRY(h, hipMemcpyAsync(b.aud_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        h->uniform_cutoffs = true;
        for (int c = 1; c < C; c++) if (idx[(size_t)c * 2] != idx[0] || idx[(size_t)c * 2 + 1] != idx[1]) { h->uniform_cutoffs = false; break; }
        h->ctx.uniform_cutoffs = h->uniform_cutoffs ? 1 : 0;
    }
    HIP_TRY(h, hipMemcpyAsync(b.b_lpr, lpr.data(), lpr.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(b.b_lmr, lmr.data(), lmr.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(b.deemph, de.data(), de.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(b.mix, mix.data(), mix.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipStreamSynchronize(s));  // the host vectors die here
    // FMD_FLAG_FAST_MATH: the IIR runs inside k_front's tile (no state, no extra stage) unless a channel's time constant is too long
    // for its warm-up; a switch between the two forms (controls changed across ~79 us) restarts the filter from the state the other
    // form last left: a transient of a millisecond on those channels
    h->ctx.deemph_in_tile = (h->ctx.fast && any && !slow_pole) ? 1 : 0;
    if (h->ctx.deemph_in_tile) { h->ctx.any_deemph = 0; h->deemph_linger = false; h->deemph_on = false; }
    else {
        if (any) { h->ctx.any_deemph = 1; h->deemph_linger = false; }
        else if (h->deemph_on) { h->ctx.any_deemph = 1; h->deemph_linger = true; }
        else if (!h->deemph_linger) h->ctx.any_deemph = 0;
        h->deemph_on = any != 0;
    }
    h->controls_dirty = false;
    return FMD_OK;
}

// Tables of the parallel form of the pilot peak filter (fmd_kernels.h PilotFastTab), in double precision.
void design_pilot_fast(const fmd_coeffs& k, PilotFastTab* t) {
    const double a0 = k.pilot_a[0], a1 = k.pilot_a[1];
    struct M2 { double a, b, c, d; };
    auto mul = [](const M2& x, const M2& y) { return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; };
    const M2 A{a1, a0, 1.0, 0.0};
    M2 P = A;                                   // A^(k+1)
    for (int i = 0; i < kPilotSeg; i++) { t->h1[i] = (float)P.a; t->h2[i] = (float)P.b; if (i + 1 < kPilotSeg) P = mul(A, P); }
    M2 S = P;                                   // M = A^kPilotSeg
    for (int s = 0; s < 4; s++) { t->m[s][0] = (float)S.a; t->m[s][1] = (float)S.b; t->m[s][2] = (float)S.c; t->m[s][3] = (float)S.d; S = mul(S, S); }
    M2 L{1.0, 0.0, 0.0, 1.0};
    for (int l = 0; l <= 32; l++) { t->mlane[l][0] = (float)L.a; t->mlane[l][1] = (float)L.b; t->mlane[l][2] = (float)L.c; t->mlane[l][3] = (float)L.d; L = mul(P, L); }
    t->k = k.pilot_b[0]; t->a0 = k.pilot_a[0]; t->a1 = k.pilot_a[1];
}

// Tables of k_pll_span (fmd_kernels.h PllSpanTab): the pilot PLL's loop filter, integrator and NCO over one span as a linear map,
// in double.  Unknowns v = (lpf, I, e1, e2, r0, eh[0..L-1]); conventions of the reference loop (broadcast_fm_demod.cpp:430-456):
// at sample n it uses the previous sample's error, lpf[n] = b0 e[n-2] + b1 e[n-1] + a0 lpf[n-1], I[n] = I[n-1] + 0.1 Ts e[n-1],
// f[n] = -19000 - 100 (0.01 lpf[n] + I[n]), t[n] = t[n-1] + Ts f[n]; with the hold at F0 = f[0] + r0 the error is
// e[n] = 2 pi (eh[n] + dev[n]), dev[n] = Ts sum_{j <= n} (f[j] - F0): substituting sample by sample is the triangular solve.
// rows[r][i]: coefficient of unknown i = (lpf, I, e1, e2, r0, eh[0..L-1]) in row r, in double
void span_rows(const fmd_coeffs& k, std::vector<double> (&rows)[kSpanRows]) {
    constexpr int L = kSpan, NV = 5 + L;
    const double b0 = k.pll_lpf_b[0], b1 = k.pll_lpf_b[1], a0 = k.pll_lpf_a[0];
    const float Ts32 = 1.0f / 128000.0f;                       // the reference's PLL_Mixer KTs (broadcast_fm_demod.cpp:226-235), a float
    const double Ts = (double)Ts32, ktsi = (double)(0.1f * Ts32), two_pi = 6.283185307179586476925;
    using Vec = std::vector<double>;
    auto unit = [&](int i) { Vec v(NV, 0.0); v[(size_t)i] = 1.0; return v; };
    Vec lpf = unit(0), I = unit(1), e1 = unit(2), e2 = unit(3), dev(NV, 0.0), g0;
    for (int n = 0; n < L; n++) {
        Vec g(NV), e(NV);
        for (int i = 0; i < NV; i++) {
            lpf[i] = b0 * e2[i] + b1 * e1[i] + a0 * lpf[i];
            I[i] += ktsi * e1[i];
            g[i] = -100.0 * (0.01 * lpf[i] + I[i]);
        }
        if (n == 0) g0 = g;
        for (int i = 0; i < NV; i++) dev[i] += Ts * (g[i] - g0[i]);
        dev[4] -= Ts;                                           // the hold runs r0 faster than f[0]
        for (int i = 0; i < NV; i++) e[i] = two_pi * dev[i];
        e[5 + n] += two_pi;
        if (n == kSpanN1) rows[2] = dev;
        if (n == kSpanN2) rows[3] = dev;
        if (n == L - 1) { rows[0] = lpf; rows[1] = I; rows[4] = dev; }
        e2 = e1; e1 = e;
    }
}

// (alpha, beta, gamma) of dev(n) ~ alpha n + beta n^2 + gamma n^3 through the three deviation rows
void span_cubic_inverse(float (&minv)[3][4]) {
    constexpr int L = kSpan;
    const double x[3] = {(double)kSpanN1, (double)kSpanN2, (double)(L - 1)};
    double A[3][3], inv[3][3];
    for (int i = 0; i < 3; i++) { A[i][0] = x[i]; A[i][1] = x[i] * x[i]; A[i][2] = x[i] * x[i] * x[i]; }
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            inv[j][i] = (A[i1][j1] * A[i2][j2] - A[i1][j2] * A[i2][j1]) / det;     // cofactor (cyclic indices carry the sign), transposed
        }
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) minv[i][j] = (float)inv[i][j]; minv[i][3] = 0.0f; }
}

void design_pll_span(const fmd_coeffs& k, PllSpanTab* t) {
    constexpr int L = kSpan;
    const float Ts32 = 1.0f / 128000.0f;
    const double Ts = (double)Ts32, two_pi = 6.283185307179586476925;
    std::vector<double> rows[kSpanRows];
    span_rows(k, rows);
    std::memset(t, 0, sizeof(*t));
    for (int r = 0; r < kSpanRows; r++) {
        for (int n = 0; n < L; n++) t->w[r][n] = (float)rows[r][(size_t)(5 + n)];
        for (int i = 0; i < 5; i++) t->s[r][i] = (float)rows[r][(size_t)i];
    }
    span_cubic_inverse(t->minv);
    // quadrature of the filtered pilot's real rail.  The reference's Hilbert rail is im[n] = sum_k b[k] s[n - 64 + k] beside
    // re[n] = s[n - 32] (hilbert_fir_filter.h:26-46); for s = cos(w0 n) that is |H(w0)| sin(w0 (n - 32)), and
    // re[n-1] - re[n+1] = 2 sin(w0) sin(w0 (n - 32)): im[n] = quad (re[n-1] - re[n+1]), quad = |H(w0)| / (2 sin w0)
    const double w0 = two_pi * 19000.0 / 128000.0;
    double hr = 0.0, hi = 0.0;
    for (int i = 0; i < 65; i++) { hr += k.b_hilbert[i] * std::cos(w0 * i); hi += k.b_hilbert[i] * std::sin(w0 * i); }
    t->quad = (float)(std::sqrt(hr * hr + hi * hi) / (2.0 * std::sin(w0)));
    t->kappa = (float)(-19000.0 * Ts + 19.0 / 128.0);
    for (int i = 0; i < 32; i++) t->hil[i] = k.b_hilbert[2 * i + 1];
}

// Tables of k_pll_sparse (fmd_kernels.h PllSparseTab; float64 model: tools/proto/sparse_pll.py design_sparse)
void design_pll_sparse(const fmd_coeffs& k, PllSparseTab* t) {
    using cd = std::complex<double>;
    constexpr int L = kSpan, D = kSparseDec, KP = kSparsePts;
    const double two_pi = 6.283185307179586476925, w0 = two_pi * 19.0 / 128.0;
    const float Ts32 = 1.0f / 128000.0f;
    std::memset(t, 0, sizeof(*t));
    // the poles of the peak filter as its float coefficients have them: y[n] = K x[n-2] + a1 y[n-1] + a0 y[n-2], a0 = -r^2, a1 = 2 r cos wp
    const double r = std::sqrt(-(double)k.pilot_a[0]), wp = std::acos((double)k.pilot_a[1] / (2.0 * r));
    const cd rho = std::polar(r, wp - w0);
    // W[q], q = -8 .. 23: the 17-tap boxcar (centred) in front of the exact decimation by 16, sum_i rho^i over the i it covers
    cd W[2 * D];
    for (int q = -8; q < 24; q++) {
        cd acc = 0.0;
        for (int i = 0; i < D; i++) if (std::abs(q - i) <= 8) acc += std::pow(rho, i);
        W[q + 8] = acc / 17.0;
    }
    cd wcd[2 * D];
    for (int tt = 0; tt < 2 * D; tt++) {                        // tap tt multiplies x[m' - 25 + tt], q = 23 - tt; the mixer relative to the point (m' = span + 16 k - 23)
        wcd[tt] = W[(23 - tt) + 8] * std::polar(1.0, -w0 * (double)(tt - 46));
        t->wre[tt] = (float)wcd[tt].real(); t->wim[tt] = (float)wcd[tt].imag();
    }
    const cd rho16 = std::pow(rho, D);
    for (int kk = 0; kk < KP; kk++) {
        const cd ro = std::polar(1.0, -w0 * (double)(D * kk)), ca = std::pow(rho16, kk + 1);
        t->rot[kk][0] = (float)ro.real(); t->rot[kk][1] = (float)ro.imag();
        t->carry[kk][0] = (float)ca.real(); t->carry[kk][1] = (float)ca.imag();
        t->nk1[kk] = (float)(D * kk + 10);
    }
    double nbar = 0.0, s2 = 0.0;
    for (int kk = 0; kk < KP; kk++) nbar += (double)(D * kk + 9) / KP;
    for (int kk = 0; kk < KP; kk++) { const double c = (double)(D * kk + 9) - nbar; t->ck[kk] = (float)c; s2 += c * c; }
    for (int s_ = 0; s_ < 3; s_++) { const cd p = std::pow(rho16, 1 << s_); t->scan[s_][0] = (float)p.real(); t->scan[s_][1] = (float)p.imag(); }
    const cd cA = cd(0.0, -1.0) * ((double)k.pilot_b[0] / std::sin(wp)) * std::polar(1.0, wp);
    double phi0 = std::arg(cA) / two_pi - 19.0 * 33.0 / 128.0;
    phi0 -= std::nearbyint(phi0);
    t->phi0 = (float)phi0; t->inv_s2 = (float)(1.0 / s2); t->nbar = (float)nbar;
    t->kappa = (float)(-19000.0 * (double)Ts32 + 19.0 / 128.0);
    double hr = 0.0, hi = 0.0;
    for (int i = 0; i < 65; i++) { hr += k.b_hilbert[i] * std::cos(w0 * i); hi += k.b_hilbert[i] * std::sin(w0 * i); }
    const double g2 = hr * hr + hi * hi;                        // |H_hilbert(w0)|^2: the reference's imaginary rail carries it
    t->pw_scale = (float)(std::norm(cA) * (1.0 + g2) * 0.5 * (double)D);
    // the non-resonant branch: Z_eff = Z - e^{-2 j wp} u_slow / (1 - rho2), rho2 = r e^{-j (wp + w0)}, u_slow = V / (DC gain of a point's 32 weights)
    cd wdc = 0.0;
    for (int q = 0; q < 2 * D; q++) wdc += W[q];
    const cd kap2 = -std::polar(1.0, -2.0 * wp) / (wdc * (1.0 - std::polar(r, -(wp + w0))));
    t->kap2[0] = (float)kap2.real(); t->kap2[1] = (float)kap2.imag();
    std::vector<double> rows[kSpanRows];
    span_rows(k, rows);
    {   // rows 2..4 -> (alpha, beta, gamma) of the deviation's cubic
        float mi[3][4];
        span_cubic_inverse(mi);
        const std::vector<double> d1 = rows[2], d2 = rows[3], d3 = rows[4];
        for (int i = 0; i < 3; i++)
            for (size_t j = 0; j < d1.size(); j++) rows[2 + i][j] = (double)mi[i][0] * d1[j] + (double)mi[i][1] * d2[j] + (double)mi[i][2] * d3[j];
    }
    for (int rr = 0; rr < kSpanRows; rr++) {
        double ws = 0.0, wm = 0.0, suf = 0.0;
        for (int n = L - 1; n >= 0; n--) {
            const double w = rows[rr][(size_t)(5 + n)];
            ws += w; wm += w * ((double)n - nbar); suf += w;
            t->sw[rr][n] = (float)suf;
        }
        t->wsum[rr] = (float)ws; t->wmom[rr] = (float)wm;
        for (int i = 0; i < 5; i++) t->s[rr][i] = (float)rows[rr][(size_t)i];
    }
}

// PllSparseTab::wrap_tie (fmd_kernels.hip wrap_tie_u8): for every u8 sample (x, y) the sign of the reference's wrapped phase difference to a
// sample in exactly the opposite direction — fm_demod.cpp:36-43 on glibc's atan2f, restated bit for bit (fmd_math.h fmd_atan2f_full)
void design_wrap_tie(uint32_t* bits2048) {
    std::memset(bits2048, 0, sizeof(uint32_t) * 2048);
    const float pi = fmd::bits_f32(fmd::kPiBits), two_pi = fmd::bits_f32(fmd::kTwoPiBits);
    for (int yr = 0; yr < 256; yr++)
        for (int xr = 0; xr < 256; xr++) {
            const float x = (float)xr - 127.0f, y = (float)yr - 127.0f;
            if (x == 0.0f && y == 0.0f) continue;
            float dl = fmd::fmd_atan2f_full(0.0f - y, 0.0f - x) - fmd::fmd_atan2f_full(y, x);       // (0 - v: the opposite sample is (float)u8 - 127 too, never -0)
            if (dl >= pi) dl = dl - two_pi;
            else if (dl <= -pi) dl = dl + two_pi;
            const unsigned key = ((unsigned)yr << 8) | (unsigned)xr;
            if (dl > 0.0f) bits2048[key >> 5] |= 1u << (key & 31u);
        }
}

void design_front_mfma(const fmd_coeffs& k, int m, std::vector<uint16_t>& img) {
    const int ks_pre = m > 1 ? (64 + (8 - m) + 15 * m + 31) / 32 : 0;      // k_predecim_mfma (PredecimGeomM::KS, ::SH)
    img.assign((size_t)kFrontImgU4 * 8 + (size_t)ks_pre * 2 * 64 * 8, 0);
    toeplitz_image(k.b_fm_out, 64, 2, 3, img.data());
    toeplitz_image(k.b_hilbert, 65, 1, 3, img.data() + (size_t)3 * 2 * 64 * 8);
    if (m > 1) toeplitz_image(k.b_fm_in, 64, m, ks_pre, img.data() + (size_t)kFrontImgU4 * 8, 8 - m);
}

void fill_ctx_coeffs(fmd_handle h) {
    const fmd_coeffs& k = h->base;
    LaunchCtx& x = h->ctx;
    std::memcpy(x.front.b_fm_in, k.b_fm_in, sizeof(k.b_fm_in));
    std::memcpy(x.front.b_fm_out, k.b_fm_out, sizeof(k.b_fm_out));
    for (int i = 0; i < 32; i++) x.front.b_hilbert_odd[i] = k.b_hilbert[2 * i + 1];
    x.front.fm_gain = k.fm_gain;
    std::memcpy(x.rds_taps.b, k.b_rds, sizeof(k.b_rds));
    x.loops.pilot_k = k.pilot_b[0]; x.loops.pilot_a0 = k.pilot_a[0]; x.loops.pilot_a1 = k.pilot_a[1];
    x.loops.pll_b0 = k.pll_lpf_b[0]; x.loops.pll_b1 = k.pll_lpf_b[1]; x.loops.pll_a0 = k.pll_lpf_a[0];
    x.loops.ted_b0 = k.ted_lpf_b[0]; x.loops.ted_b1 = k.ted_lpf_b[1]; x.loops.ted_a0 = k.ted_lpf_a[0];
    x.loops.bpsk_b0 = k.bpsk_lpf_b[0]; x.loops.bpsk_b1 = k.bpsk_lpf_b[1]; x.loops.bpsk_a0 = k.bpsk_lpf_a[0];
}

int zero_history(fmd_handle h, hipStream_t s) {
    const Dims& d = h->ctx.d;
    Buffers& b = h->ctx.b;
    for (int p = 0; p < 2; p++) {
        HIP_TRY(h, hipMemsetAsync(b.base_tail[p], 0, sizeof(float2) * (size_t)d.C * d.tail_base, s));
        if (b.pre_tail[p]) HIP_TRY(h, hipMemsetAsync(b.pre_tail[p], 0, sizeof(float2) * (size_t)d.C * 64, s));
        HIP_TRY(h, hipMemsetAsync(b.iq_tail[p], 0, sizeof(float2) * (size_t)d.C * 128, s));
        HIP_TRY(h, hipMemsetAsync(b.dt_tail[p], 0, sizeof(float) * (size_t)d.C * 128, s));
        HIP_TRY(h, hipMemsetAsync(b.fo_tail[p], 0, sizeof(float) * (size_t)d.C * 64, s));
        HIP_TRY(h, hipMemsetAsync(b.lmr_est[p], 0, sizeof(float) * (size_t)d.C * d.n_est, s));
        if (b.pv_hist[p]) HIP_TRY(h, hipMemsetAsync(b.pv_hist[p], 0, sizeof(float4) * (size_t)d.C * 4, s));
    }
    for (int p = 0; p < kSlots; p++) {
        if (b.fo_pl[p]) {   // the history in front of the planes' rows (and the rows themselves)
            HIP_TRY(h, hipMemsetAsync(b.fo_pl[p], 0, sizeof(float) * (size_t)d.C * ((size_t)kFoPad + d.n_fm_out), s));
            HIP_TRY(h, hipMemsetAsync(b.pll_poly[p], 0, sizeof(float4) * (size_t)d.C * ((size_t)1 + d.n_fm_out / kSpan), s));
        }
        HIP_TRY(h, hipMemsetAsync(b.rds_count[p], 0, sizeof(int) * (size_t)d.C, s));
        HIP_TRY(h, hipMemsetAsync(b.rds_bytes_count[p], 0, sizeof(int) * (size_t)d.C, s));
    }
    HIP_TRY(h, launch_reset_state(h->ctx, s));
    // everything is idle here (callers synchronise first): restart the per-wavefront PLL hand-over chain, watchdog flag included
    if (b.pll_chain) HIP_TRY(h, hipMemsetAsync(b.pll_chain, 0, sizeof(unsigned) * ((size_t)h->pll_waves + 1 + (size_t)d.C + 2), s));      // (and the body hints behind it)
    if (h->pll_unl_host) h->pll_unl_host[0] = h->pll_unl_host[1] = 0u;
    h->ctx.pll_unlocked_now = false; h->ctx.pll_launch_no = 0;
    h->pll_seq = 0;
    h->n_blocks = 0;
    h->warm_left = h->ctx.fast ? (int)((8192 + d.n_fm_out - 1) / d.n_fm_out) : 0;     // kPllWarmSamples of every station's life
    if (h->ctx.fast && dev_env("FMD_DEBUG_PLL_DENSE")) h->warm_left = 1 << 30;   // development knob: k_pll_span for every block (A/B against round 3's pilot stage)
    h->deferred.active = false; h->last_x_event = nullptr; h->last_p_event = nullptr;
    for (hipEvent_t& e : h->x_done) e = nullptr;
    h->ev_consumed = nullptr;
    h->out_slot = 0; h->sub_slot = 0; h->have_out = false; h->out_block = -1;
    h->chain_blocks = 0; h->last_block_chain = false;
    h->epoch++;
    for (bool& u : h->slot_used) u = false;
    for (bool& u : h->consumer_pending) u = false;
    h->poisoned = false;
    return FMD_OK;
}

// The extract + RDS stages of the block whose launch fmd_submit_* put off (tolerance mode).
// behind_front (the next block has just been submitted): k_extract_bp goes on the FRONT END's stream.  The two throughput kernels
// gain nothing from running side by side — together they took longer than one after the other (tools/r3_timeline.sh: 0.37 ms a block for
// the pair against 0.14 + 0.17 ms alone; they share a CU's LDS and wave slots, and every hop between queues costs ~50 us) — so they take
// turns on one queue, in the order front(k + 1), extract(k), front(k + 2), ...; by the time extract(k) is reached, the pilot loop of
// block k has run on its own queue, as the RDS stages do.  Otherwise (somebody asks for the block's outputs before the next block is
// there: fmd_wait_outputs, a getter, fmd_synchronize) it goes on the extract stream as in the exact mode, beside the next front end.
// The pilot stage of the put-off block as a launch of its own on `sP`: the front end's queue, or the pilot queue as in the undeferred
// arrangement (consecutive blocks' pilot stages must run in order: a block whose stage is queued at submission sends the put-off one ahead).
int launch_deferred_pll(fmd_handle h, hipStream_t sP) {
    auto& q = h->deferred;
    if (!q.active || !q.pll_pending) return FMD_OK;
    q.pll_pending = false;
    if (h->last_p_event && h->last_p_stream != sP) HIP_TRY(h, hipStreamWaitEvent(sP, h->last_p_event, 0));
    if (sP != h->sF || q.front_cross) HIP_TRY(h, hipStreamWaitEvent(sP, q.front_dep, 0));
    if (sP != h->sF) {          // (see process_dev: the stage also writes the history in front of the next slot's rows)
        const int nx = (q.slot + 1) % kSlots;
        if (h->slot_used[nx] && h->x_done[nx]) HIP_TRY(h, hipStreamWaitEvent(sP, h->x_done[nx], 0));
    }
    SlotRef r = q.ref;
    if (q.pm && q.prof_p) { r.t0 = q.pm->t0[ST_PLL]; r.t1 = q.pm->t1[ST_PLL]; q.pm->used[ST_PLL] = true; }
    if (!r.t1) r.done = h->ev_B[q.slot];
    q.pll_dep = r.t1 ? r.t1 : h->ev_B[q.slot];
    q.pll_stream = sP;
    hipError_t e = (h->debug_skip & (1u << ST_PLL)) ? hipEventRecord(q.pll_dep, sP) : launch_stage_pll(h->ctx, r, sP);
    if (e != hipSuccess) { h->poisoned = true; return fail(h, FMD_ERR_DEVICE, "k_pll_sparse launch: %s", hipGetErrorString(e)); }
    h->last_p_stream = sP; h->last_p_event = q.pll_dep;
    return FMD_OK;
}

int launch_deferred(fmd_handle h, bool behind_front) {
    auto& q = h->deferred;
    if (!q.active) return FMD_OK;
    hipStream_t sXq = (behind_front && !h->split_queues) ? h->sF : h->sX, sR = h->sR;
    if (h->consumer_pending[q.slot]) {       // fmd_release_outputs: a consumer still reads this slot's old outputs
        HIP_TRY(h, hipStreamWaitEvent(sXq, h->ev_C[q.slot], 0));
        HIP_TRY(h, hipStreamWaitEvent(sR, h->ev_C[q.slot], 0));
        h->consumer_pending[q.slot] = false;
    }
    if (h->last_x_event && h->last_x_stream != sXq) HIP_TRY(h, hipStreamWaitEvent(sXq, h->last_x_event, 0));
    { int rc = launch_deferred_pll(h, behind_front ? h->sF : h->sB); if (rc) return rc; }
    q.active = false;
    if (q.pll_dep && q.pll_stream != sXq) HIP_TRY(h, hipStreamWaitEvent(sXq, q.pll_dep, 0));
    hipEvent_t dep;
    {
        SlotRef r = q.ref;
        if (q.pm && q.prof_x) { r.t0 = q.pm->t0[ST_EXTRACT]; r.t1 = q.pm->t1[ST_EXTRACT]; q.pm->used[ST_EXTRACT] = true; }
        if (!r.t1) r.done = h->ev_E[q.slot];
        dep = r.t1 ? r.t1 : h->ev_E[q.slot];
        hipError_t e = (h->debug_skip & (1u << ST_EXTRACT)) ? hipEventRecord(dep, sXq) : launch_stage_extract(h->ctx, r, sXq);
        if (e != hipSuccess) { h->poisoned = true; return fail(h, FMD_ERR_DEVICE, "k_extract launch: %s", hipGetErrorString(e)); }
        h->last_x_stream = sXq; h->last_x_event = dep; h->x_done[q.slot] = dep;
    }
    HIP_TRY(h, hipStreamWaitEvent(sR, dep, 0));
    {
        SlotRef r = q.ref;
        if (q.pm && q.prof_r) { r.t0 = q.pm->t0[ST_RDS]; r.t1 = q.pm->t1[ST_RDS]; q.pm->used[ST_RDS] = true; }
        if (!r.t1) r.done = h->ev_X[q.slot];
        dep = r.t1 ? r.t1 : h->ev_X[q.slot];
        hipError_t e = (h->debug_skip & (1u << ST_RDS)) ? hipEventRecord(dep, sR) : launch_stage_rds(h->ctx, r, sR);
        if (e != hipSuccess) { h->poisoned = true; return fail(h, FMD_ERR_DEVICE, "k_rds_sync launch: %s", hipGetErrorString(e)); }
    }
    if (dep != h->ev_X[q.slot]) HIP_TRY(h, hipEventRecord(h->ev_X[q.slot], sR));
    h->out_slot = q.slot; h->have_out = true; h->out_block = q.block;
    return FMD_OK;
}

// A caller is about to use the device views of the newest outputs (fmd_wait_outputs, fmd_release_outputs, the *_dev getters).
// Default: they are the newest BLOCK's — if its extract stage is still put off it is queued now, on the extract stream, and from here on
// the handle queues every block's stages at once (such a caller asks after every block: taking turns on the front end's queue would
// stall that queue for the length of the pilot loop each time).  fmd_set_output_lag: nothing is forced, the views are the newest queued ones.
int outputs_wanted(fmd_handle h) {
    if (h->lag_outputs || !h->deferred.active) return FMD_OK;
    HIP_TRY(h, hipSetDevice(h->device));        // (the *_dev getters reach this without one: launches follow)
    h->lazy_extract = false;
    return launch_deferred(h, false);
}

int sync_all(fmd_handle h) {
    HIP_TRY(h, hipSetDevice(h->device));
    { int rc = launch_deferred(h, true); if (rc) return rc; }    // (everything drains: the front end's queue is as good as any)
    for (hipStream_t st : {h->sF, h->sD, h->sA, h->sB, h->sB2, h->sX, h->sR, h->own_stream}) if (st) HIP_TRY(h, hipStreamSynchronize(st));
    h->last_x_event = nullptr;        // (everything has run: no order left to keep; a timed block's events are about to be freed)
    h->last_p_event = nullptr;
    for (hipEvent_t& e : h->x_done) e = nullptr;
    if (!h->pipelined && h->n_blocks > 0) HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    if (h->pll_chained && h->pll_seq) {   // the hand-over watchdog of k_pilot_pll
        unsigned timed_out = 0;
        HIP_TRY(h, hipMemcpy(&timed_out, h->ctx.b.pll_chain + h->pll_waves, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (timed_out) {
            // reported once; the flag is cleared so that later calls do not fail for good, and the handle asks for a reset
            (void)hipMemset(h->ctx.b.pll_chain + h->pll_waves, 0, sizeof(unsigned));
            h->poisoned = true;
            return fail(h, FMD_ERR_DEVICE, "k_pilot_pll: a wavefront's predecessor never published its state (call fmd_reset)");
        }
    }
    return FMD_OK;
}

// ordered: fmd_process_*_dev (the caller's stream is ordered behind the library's read of the block); otherwise fmd_submit_*_dev
// (`stream` only says when the input is ready, NULL = now; nothing is queued on it)
template <typename InT>
int process_dev(fmd_handle h, const InT* d_iq, int n_channels, int n_samples, void* stream, bool ordered = true) {
    if (!h) return FMD_ERR_ARG;
    if (!d_iq) return fail(h, FMD_ERR_ARG, "null input pointer");
    if (n_channels != h->cfg.n_channels || n_samples != h->cfg.block_size)
        return fail(h, FMD_ERR_SIZE, "block dropped: got %d x %d, handle is %d x %d", n_channels, n_samples, h->cfg.n_channels, h->cfg.block_size);
    if (h->poisoned) return fail(h, FMD_ERR_STATE, "an earlier block failed part-way: call fmd_reset");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->controls_dirty) {
        int rc = sync_all(h);          // control tables are read by in-flight stages: drain before rewriting them
        if (!rc) rc = upload_controls(h, h->own_stream);
        if (rc) return rc;
    }
    const int slot = (int)(h->n_blocks % kSlots);
    SlotRef ref{slot, (int)(h->n_blocks & 1), nullptr, nullptr};
    ref.warm = (h->ctx.fast && h->warm_left > 0) ? (h->warm_left >= (1 << 29) ? 2 : 1) : 0;
    const bool u8 = sizeof(InT) == 2;
    const bool pipe = h->pipelined;
    if (!pipe) ordered = true;                     // every stage runs on `s` itself
    hipStream_t sF = pipe ? h->sF : s, sA = pipe ? h->sA : s, sX = pipe ? h->sX : s, sR = pipe ? h->sR : s;
    const bool lazy = pipe && h->lazy_extract && !ordered;
    // consecutive blocks' PLL launches alternate between two streams when they hand over per wavefront (fmd_kernels.hip)
    const bool chained = pipe && h->pll_chained;
    hipStream_t sB = pipe ? ((chained && (h->n_blocks & 1)) ? h->sB2 : h->sB) : s;
    ProfiledBlock* pm = nullptr;
    if (h->profiling) {
        pm = new ProfiledBlock();
        for (int i = 0; i < ST_COUNT; i++) { pm->used[i] = false; HIP_TRY(h, hipEventCreate(&pm->t0[i])); HIP_TRY(h, hipEventCreate(&pm->t1[i])); }
        h->marks.push_back(pm);
    }
    // event bracketing perturbs the pipeline it measures (extra queue packets between dependent kernels): mode 2 keeps it
    // on the dominant kernel and samples the others
    // mode 3 samples: every stage of every 4th block, plus the PLL stage of the block behind it (for the hand-over gap)
    auto prof_stage = [&](int st) {
        if (h->profiling == 3) return (h->n_blocks & 3) == 0 || (st == ST_PLL && (h->n_blocks & 3) == 1);
        return h->profiling == 1 || st == ST_PLL || (h->n_blocks & 3) == 0;
    };
    // The event that orders the next stage behind this one rides on the stage's last dispatch packet (SlotRef::done).  A
    // stage that is being timed already carries its stop event there: the next stage then waits on that one (dep).
    hipEvent_t dep = nullptr;
    auto run = [&](Stage st, hipStream_t on, hipError_t (*fn)(const LaunchCtx&, SlotRef, hipStream_t), hipEvent_t done) -> hipError_t {
        SlotRef r = ref;
        if (pm && prof_stage(st)) { r.t0 = pm->t0[st]; r.t1 = pm->t1[st]; pm->used[st] = true; }
        if (pipe && !r.t1) r.done = done;
        if (st == ST_PLL && chained) r.seq = ++h->pll_seq;
        dep = r.t1 ? r.t1 : done;
        if (h->debug_skip & (1u << st)) return pipe ? hipEventRecord(dep, on) : hipSuccess;
        return fn(h->ctx, r, on);
    };
    hipError_t e = hipSuccess;
    // from here on kernels are queued and host-side counters advance: a failure leaves the state between two blocks
    struct Poison { fmd_handle h; bool armed = true; ~Poison() { if (armed) h->poisoned = true; } } poison{h};
    // fmd_release_outputs: the writers of this slot's output views wait for the consumer that still reads the old contents
    if (h->consumer_pending[slot] && !lazy) {
        HIP_TRY(h, hipStreamWaitEvent(sX, h->ev_C[slot], 0));
        if (sR != sX) HIP_TRY(h, hipStreamWaitEvent(sR, h->ev_C[slot], 0));
        h->consumer_pending[slot] = false;
    }
    // The first decimator (1.024 / 2.048 MSa/s) gets a stream of its own when the PLL launches do not need own_stream: it then
    // works on block b+1 while k_front works on block b (back to back on one stream the two were the longest stage)
    const bool fused_pre = front_takes_capture(h->ctx);      // (tolerance mode: one kernel, k_front_pre_mfma)
    const bool predecim = h->ctx.d.m > 1 && !fused_pre;
    hipStream_t sP = (pipe && h->ctx.d.m > 1 && !chained) ? h->own_stream : sF;
    // Deferred schedule at 1.024 / 2.048 MSa/s: the front end (with the previous block's pilot stage riding it) follows the first
    // decimator on that queue, and the extract stages have the front end's queue to themselves: two queues that each run ahead,
    // instead of one on which k_extract_bp and the front end take turns while the decimator works beside both.
    hipStream_t sFq = (lazy && h->ctx.d.m > 1 && sP != sF && h->ctx.fast && h->front_with_predecim) ? sP : sF;
    hipStream_t s_first = predecim ? sP : sFq;     // the stream of the stage that reads the caller's input
    if (pipe) {
        // input is ready once everything queued so far on the caller's stream has run
        if (ordered || s) {
            HIP_TRY(h, hipEventRecord(h->ev_in, s));
            HIP_TRY(h, hipStreamWaitEvent(s_first, h->ev_in, 0));
        }
        // WAR: this slot's fm_in / fm_out_iq / pilot / pll_dt were last read by the stages of the block kSlots blocks ago
        if (h->slot_used[slot]) {
            HIP_TRY(h, hipStreamWaitEvent(sF, h->ev_X[slot], 0));
            if (sP != sF) HIP_TRY(h, hipStreamWaitEvent(sP, h->ev_X[slot], 0));
        }
    }
    // the block after the last de-emphasised one: k_front maintains the Hilbert history (fo_tail) again and must not overwrite
    // what the previous block's k_hilbert, on its own stream, is still reading
    if (pipe && !h->ctx.any_deemph && h->last_block_deemph) HIP_TRY(h, hipStreamWaitEvent(sFq, h->ev_F[h->sub_slot], 0));
    // Tolerance mode, 256 kSa/s cf32, a steady block: front end, pilot stage and extract stage as ONE launch (k_chain, fmd_kernels_chain.inc) on the
    // front end's queue, the RDS stage behind it on its own.  Nothing of such a block is put off: its outputs are complete one launch and the
    // RDS stage after its submission, and no kernel of it waits for another queue.  Start-up blocks, de-emphasised or differently filtered
    // stations, u8 captures, FMD_FLAG_KEEP_TAPS and un-pipelined handles keep the three launches; both forms leave the same histories.
    const bool chain = pipe && !u8 && !ref.warm && !h->chain_off && h->uniform_cutoffs && !h->ctx.any_deemph && !h->ctx.deemph_in_tile && !h->last_block_deemph &&
                       !(h->debug_skip & ~(1u << ST_RDS)) && chain_possible(h->ctx);
    if (chain) {
        { int rc = launch_deferred(h, true); if (rc) return rc; }       // (a start-up block ahead of this one: its put-off stages first, in order on this queue)
        if (h->consumer_pending[slot]) {
            HIP_TRY(h, hipStreamWaitEvent(sF, h->ev_C[slot], 0));
            HIP_TRY(h, hipStreamWaitEvent(sR, h->ev_C[slot], 0));
            h->consumer_pending[slot] = false;
        }
        const int nx = (slot + 1) % kSlots;      // (the block writes the histories in front of the next slot's rows: see the pilot stage below)
        if (h->slot_used[nx] && h->x_done[nx]) HIP_TRY(h, hipStreamWaitEvent(sF, h->x_done[nx], 0));      // (whichever queue that extract stage ran on)
        if (h->last_p_event && h->last_p_stream != sF) HIP_TRY(h, hipStreamWaitEvent(sF, h->last_p_event, 0));
        if (h->last_x_event && h->last_x_stream != sF) HIP_TRY(h, hipStreamWaitEvent(sF, h->last_x_event, 0));
        SlotRef r = ref;
        if (pm && prof_stage(ST_FRONT)) { r.t0 = pm->t0[ST_FRONT]; r.t1 = pm->t1[ST_FRONT]; pm->used[ST_FRONT] = true; pm->chain = true; }
        if (!r.t1) r.done = h->ev_E[slot];
        dep = r.t1 ? r.t1 : h->ev_E[slot];
        e = launch_stage_chain(h->ctx, r, d_iq, sF);
        if (e != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_chain launch: %s", hipGetErrorString(e));
        if (ordered) HIP_TRY(h, hipStreamWaitEvent(s, dep, 0));          // the caller may reuse `iq` in stream order after this call
        HIP_TRY(h, hipEventRecord(h->ev_F[slot], sF));                   // fmd_wait_input: an event that outlives this call
        h->ev_consumed = h->ev_F[slot];
        h->last_p_stream = sF; h->last_p_event = dep; h->last_x_stream = sF; h->last_x_event = dep; h->x_done[slot] = dep;
        HIP_TRY(h, hipStreamWaitEvent(sR, dep, 0));
        if ((e = run(ST_RDS, sR, launch_stage_rds, h->ev_X[slot])) != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_rds_sync launch: %s", hipGetErrorString(e));
        if (dep != h->ev_X[slot]) HIP_TRY(h, hipEventRecord(h->ev_X[slot], sR));
        h->last_block_deemph = false;
        h->slot_used[slot] = true;
        h->sub_slot = slot;
        h->out_slot = slot; h->have_out = true; h->out_block = h->n_blocks;
        h->n_blocks++;
        h->chain_blocks++; h->last_block_chain = true;
        h->last_stream = s;
        poison.armed = false;
        return FMD_OK;
    }
    h->last_block_chain = false;
    hipEvent_t input_done = nullptr;               // fires when the caller's buffer has been consumed
    if (predecim) {
        SlotRef r = ref;
        if (pm && prof_stage(ST_PREDECIM)) { r.t0 = pm->t0[ST_PREDECIM]; r.t1 = pm->t1[ST_PREDECIM]; pm->used[ST_PREDECIM] = true; }
        if (pipe && !r.t1) r.done = h->ev_P[slot];
        input_done = r.t1 ? r.t1 : h->ev_P[slot];
        e = launch_stage_predecim(h->ctx, r, d_iq, u8, sP);
        if (e != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_predecim launch: %s", hipGetErrorString(e));
        if (pipe && sP != sFq) HIP_TRY(h, hipStreamWaitEvent(sFq, input_done, 0));
    }
    {
        SlotRef r = ref;
        if (pm && prof_stage(ST_FRONT)) { r.t0 = pm->t0[ST_FRONT]; r.t1 = pm->t1[ST_FRONT]; pm->used[ST_FRONT] = true; }
        hipEvent_t front_done = h->ctx.any_deemph ? h->ev_D[slot] : h->ev_F[slot];
        if (pipe && !r.t1) r.done = front_done;
        dep = r.t1 ? r.t1 : front_done;
        // the previous block's pilot stage, put off with its extract stage: as the first workgroups of this launch
        const SlotRef* ride = nullptr;
        auto& q = h->deferred;
        if (lazy && q.active && q.pll_pending && !q.ref.warm && !q.deemph && q.front_stream == sFq && !h->ctx.any_deemph && !h->ctx.b.fm_out_iq[q.slot] &&
            !(h->debug_skip & ((1u << ST_PLL) | (1u << ST_FRONT))) && !h->no_fused_pll) {
            ride = &q.ref;
            q.pll_pending = false; q.pll_dep = (sFq != sF || h->split_queues) ? dep : nullptr; q.pll_stream = sFq;
            if (h->last_p_event && h->last_p_stream != sFq) HIP_TRY(h, hipStreamWaitEvent(sFq, h->last_p_event, 0));
            h->last_p_stream = sFq; h->last_p_event = dep;          // (the launch's own event)
        }
        if (h->debug_skip & (1u << ST_FRONT)) e = pipe ? hipEventRecord(dep, sFq) : hipSuccess;
        else e = launch_stage_front(h->ctx, r, d_iq, u8, sFq, ride);
    }
    if (e != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_front launch: %s", hipGetErrorString(e));
    hipEvent_t front_dep = dep;                    // k_front itself: the caller's buffer (256 kSa/s captures) has been consumed
    if (h->ctx.any_deemph) {
        // the optional de-emphasis IIR + Hilbert FIR: a pipeline stage of its own (stream sD), so that k_front of the next block
        // runs beside it — in k_front's stream the two made the front end the longest stage (+25 % on the step)
        hipStream_t sDe = pipe ? h->sD : s;
        if (pipe) HIP_TRY(h, hipStreamWaitEvent(sDe, dep, 0));
        if ((e = run(ST_DEEMPH, sDe, launch_stage_deemph, h->ev_F[slot])) != hipSuccess)
            return fail(h, FMD_ERR_DEVICE, "de-emphasis launch: %s", hipGetErrorString(e));
    }
    if (pipe && ordered) HIP_TRY(h, hipStreamWaitEvent(s, input_done ? input_done : front_dep, 0));   // the caller may reuse `iq` in stream order after this call
    if (pipe) {
        // fmd_wait_input: an event that outlives this call (a timed stage's stop event belongs to the profiling marks)
        hipEvent_t persistent = predecim ? h->ev_P[slot] : (h->ctx.any_deemph ? h->ev_D[slot] : h->ev_F[slot]);
        if ((input_done ? input_done : front_dep) != persistent) HIP_TRY(h, hipEventRecord(persistent, predecim ? sP : sFq));
        h->ev_consumed = persistent;
    }
    if (h->pll_k_adaptive) {
        const unsigned done = reinterpret_cast<volatile unsigned*>(h->pll_unl_host)[1], heavy = reinterpret_cast<volatile unsigned*>(h->pll_unl_host)[0];
        h->ctx.pll_unlocked_now = heavy != 0u && (heavy > done || done - heavy < 8u);      // (heavy > done: the launch that is still running has such wavefronts)
        h->ctx.pll_launch_no++;
        // (no ordering against the PLL launches: both words only grow, whichever values the copy finds will do)
        if ((h->n_blocks & 1) == 0) HIP_TRY(h, hipMemcpyAsync(h->pll_unl_host, h->ctx.b.pll_hint + h->ctx.d.C, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, sA));
    }
    if (!h->ctx.fast) {   // (FMD_FLAG_FAST_MATH: the pilot peak filter runs inside the PLL kernel, there is no power pass)
        if (pipe) HIP_TRY(h, hipStreamWaitEvent(sA, dep, 0));
        if ((e = run(ST_POWER, sA, launch_stage_power, h->ev_A[slot])) != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_pilot_power launch: %s", hipGetErrorString(e));
    }
    const hipEvent_t fm_out_dep = dep;              // behind this event the block's fm_out is complete
    const bool fm_out_cross = h->ctx.any_deemph != 0 || sFq != sF;
    const bool pll_now = !lazy || h->pll_eager;
    if (pll_now) {
    { int rc = launch_deferred_pll(h, sB); if (rc) return rc; }        // (the put-off block's pilot stage first)
    if (pipe) HIP_TRY(h, hipStreamWaitEvent(sB, dep, 0));
    // Tolerance mode: the pilot stage of block k also writes the history in front of the NEXT slot's rows (fm_out tail, last span's cubic),
    // which the extract stage of the block that last used that slot (k - 5) reads.  On the deferred schedule that stage sits ahead on
    // the front end's queue; otherwise (fmd_process_*, small batches, a consumer holding outputs back) nothing else orders the two.
    if (pipe && h->ctx.fast) {
        const int nx = (slot + 1) % kSlots;
        if (h->slot_used[nx] && h->x_done[nx]) HIP_TRY(h, hipStreamWaitEvent(sB, h->x_done[nx], 0));
    }
    if (pipe && h->ctx.fast && h->last_p_event && h->last_p_stream != sB) HIP_TRY(h, hipStreamWaitEvent(sB, h->last_p_event, 0));
    if ((e = run(ST_PLL, sB, launch_stage_pll, h->ev_B[slot])) != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_pilot_pll launch: %s", hipGetErrorString(e));
    if (pipe && h->ctx.fast) { h->last_p_stream = sB; h->last_p_event = dep; }
    }
    // the previous block's extract + RDS stages, if fmd_submit_* put them off: behind this block's front end (launch_deferred)
    { int rc = launch_deferred(h, true); if (rc) return rc; }
    if (lazy) {
        auto& q = h->deferred;
        q.active = true; q.pll_pending = !pll_now; q.front_cross = fm_out_cross; q.deemph = h->ctx.any_deemph != 0; q.front_stream = sFq; q.ref = ref; q.pll_dep = pll_now ? dep : nullptr; q.pll_stream = pll_now ? sB : nullptr; q.front_dep = fm_out_dep;
        q.slot = slot; q.block = h->n_blocks; q.pm = pm;
        q.prof_p = pm && prof_stage(ST_PLL); q.prof_x = pm && prof_stage(ST_EXTRACT); q.prof_r = pm && prof_stage(ST_RDS);
    } else {
        if (pipe) HIP_TRY(h, hipStreamWaitEvent(sX, dep, 0));
        if (pipe && h->last_x_event && h->last_x_stream != sX) HIP_TRY(h, hipStreamWaitEvent(sX, h->last_x_event, 0));
        // the extract stage's event fires behind k_extract itself: k_rds_sync does not need k_lmr_phase (same stream, behind it)
        if ((e = run(ST_EXTRACT, sX, launch_stage_extract, h->ev_E[slot])) != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_extract launch: %s", hipGetErrorString(e));
        if (pipe) { h->last_x_stream = sX; h->last_x_event = dep; h->x_done[slot] = dep; }
        if (pipe) HIP_TRY(h, hipStreamWaitEvent(sR, dep, 0));
        if ((e = run(ST_RDS, sR, launch_stage_rds, h->ev_X[slot])) != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_rds_sync launch: %s", hipGetErrorString(e));
        // ev_X outlives this call (slot reuse, fmd_wait_outputs): when the dispatch carried a timing event instead, record it
        if (pipe && dep != h->ev_X[slot]) HIP_TRY(h, hipEventRecord(h->ev_X[slot], sR));
    }
    h->last_block_deemph = h->ctx.any_deemph != 0;
    h->slot_used[slot] = true;
    h->sub_slot = slot;
    if (!lazy) { h->out_slot = slot; h->have_out = true; h->out_block = h->n_blocks; }
    h->n_blocks++;
    if (h->warm_left > 0) h->warm_left--;
    if (h->deemph_linger) { h->deemph_linger = false; h->ctx.any_deemph = 0; }
    h->last_stream = s;
    poison.armed = false;
    return FMD_OK;
}

template <typename InT>
int process_host(fmd_handle h, const InT* iq, int n_channels, int n_samples) {
    if (!h) return FMD_ERR_ARG;
    if (!iq) return fail(h, FMD_ERR_ARG, "null input pointer");
    if (n_channels != h->cfg.n_channels || n_samples != h->cfg.block_size)
        return fail(h, FMD_ERR_SIZE, "block dropped: got %d x %d, handle is %d x %d", n_channels, n_samples, h->cfg.n_channels, h->cfg.block_size);
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t bytes = sizeof(InT) * (size_t)n_channels * n_samples;
    if (h->d_in_bytes < bytes) {
        if (h->d_in) HIP_TRY(h, hipFree(h->d_in));
        h->d_in = nullptr; h->d_in_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->d_in, bytes));
        h->d_in_bytes = bytes;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_in, iq, bytes, hipMemcpyHostToDevice, h->own_stream));
    int rc = process_dev<InT>(h, static_cast<const InT*>(h->d_in), n_channels, n_samples, h->own_stream);
    if (rc) return rc;
    return sync_all(h);
}

void free_marks(fmd_handle h) {
    for (ProfiledBlock* pm : h->marks) {
        for (int i = 0; i < ST_COUNT; i++) { (void)hipEventDestroy(pm->t0[i]); (void)hipEventDestroy(pm->t1[i]); }
        delete pm;
    }
    h->marks.clear();
}

}  // namespace

extern "C" {

int fmd_api_version(void) { return FMD_API_VERSION; }
int fmd_output_lifetime_blocks(void) { static_assert(FMD_OUTPUT_LIFETIME_BLOCKS == kSlots - 1, "one contract"); return kSlots - 1; }

const char* fmd_status_string(int s) {
    switch (s) {
        case FMD_OK: return "ok";
        case FMD_ERR_ARG: return "bad argument";
        case FMD_ERR_SIZE: return "block size mismatch (block dropped)";
        case FMD_ERR_DEVICE: return "HIP runtime error";
        case FMD_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback)";
        case FMD_ERR_NAME: return "unknown stream name";
        case FMD_ERR_STATE: return "a block failed part-way: call fmd_reset";
        default: return "unknown status";
    }
}

int fmd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

void fmd_default_controls(fmd_controls* c) {
    // reference Broadcast_FM_Demod_Controls defaults (broadcast_fm_demod.h:82-88) and the SetValue() calls of the
    // constructor (broadcast_fm_demod.cpp:189,248,260)
    c->audio_out = FMD_AUDIO_STEREO;
    c->audio_stereo_mix_factor = 1.0f;
    c->use_deemphasis = 0;
    c->deemphasis_tus = 1;
    c->lpr_cutoff_hz = 15000;
    c->lmr_cutoff_hz = 15000;
}

int fmd_default_config(fmd_config* cfg, int n_channels, int fs_baseband) {
    if (!cfg || n_channels <= 0 || (fs_baseband != 256000 && fs_baseband != 1024000 && fs_baseband != 2048000)) return FMD_ERR_ARG;
    cfg->n_channels = n_channels;
    cfg->block_size = 16384 * (fs_baseband / 256000);     // 64 ms: the reference's 65536 samples at 1.024 MSa/s
    cfg->fs_baseband = fs_baseband;
    cfg->device = -1;
    cfg->flags = FMD_FLAG_FAST_MATH;
    return FMD_OK;
}

int fmd_create(const fmd_config* cfg, fmd_handle* out) {
    if (!out) return FMD_ERR_ARG;
    *out = nullptr;
    int m = 0;
    if (!config_ok(cfg, &m)) return fail(nullptr, FMD_ERR_ARG, "unsupported configuration");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, FMD_ERR_NO_DEVICE, "no HIP device");
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) return fail(nullptr, FMD_ERR_NO_DEVICE, "hipGetDevice failed"); }
    if (dev >= ndev) return fail(nullptr, FMD_ERR_ARG, "device %d out of range", dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, FMD_ERR_NO_DEVICE, "device %d is not gfx950", dev);

    fmd_handle h = new (std::nothrow) fmd_handle_s();
    if (!h) return FMD_ERR_ARG;
    h->cfg = *cfg;
    h->cfg.device = dev;
    h->device = dev;
    auto bail = [&](int rc) { g_create_error = h->err; fmd_destroy(h); return rc; };
    if (hipSetDevice(dev) != hipSuccess) return bail(fail(h, FMD_ERR_DEVICE, "hipSetDevice failed"));
    { hipError_t e = prepare_kernels(); if (e != hipSuccess) return bail(fail(h, FMD_ERR_DEVICE, "prepare_kernels: %s", hipGetErrorString(e))); }
    // development switch FMD_CU_MASK_F / FMD_CU_MASK_X (hex words, 8 x 32 bits, e.g. ffff-ffff-...): the front end's queues and the extract
    // stage's queue restricted to those CUs (tools/cumask_ab.sh)
    std::vector<uint32_t> mask_f, mask_x;
    auto parse_mask = [](const char* e, std::vector<uint32_t>& m) { while (e && *e) { char* end = nullptr; const unsigned long v = strtoul(e, &end, 16); if (!end || end == e) break; m.push_back((uint32_t)v); e = *end ? end + 1 : end; } };   // (any one non-hex character separates the words)
    parse_mask(dev_env("FMD_CU_MASK_F"), mask_f); parse_mask(dev_env("FMD_CU_MASK_X"), mask_x);
    h->split_queues = (!mask_f.empty() && !mask_x.empty()) || dev_env("FMD_SPLIT_QUEUES") != nullptr;     // FMD_SPLIT_QUEUES: the two-queue schedule on plain (unmasked) queues, with FMD_STREAM_PRIORITIES
    { hipError_t e = !mask_f.empty() ? hipExtStreamCreateWithCUMask(&h->own_stream, (uint32_t)mask_f.size(), mask_f.data()) : hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
      if (e != hipSuccess) return bail(fail(h, FMD_ERR_DEVICE, "stream: %s", hipGetErrorString(e))); }

    h->pipelined = (cfg->flags & FMD_FLAG_NO_PIPELINE) == 0;
    if (h->pipelined) warn_hw_queues_once();

    // (A CU-mask split between the serial and the FIR streams was measured: it shields the PLL wave from FIR waves
    //  sharing its SIMD — 3.06 -> 2.75 ms — but CU-masked streams did not overlap with each other on this runtime, so
    //  the step got slower overall.  Plain streams + s_setprio in the serial kernels it is.)
    // (the second PLL stream is own_stream: one more stream would be the ninth on the device with the caller's and would share a
    //  hardware queue with another stage; everything else own_stream does is preceded by a full synchronisation)
    h->sB2 = h->own_stream;
    {
        // experiment hook: FMD_STREAM_PRIORITIES="f,b,x,r" — priorities of the front / PLL / extract / RDS streams (0 = default, negative = higher)
        int prio[6] = {0, 0, 0, 0, 0, 0};       // sF, sD, sA, sB, sX, sR
        if (const char* e = dev_env("FMD_STREAM_PRIORITIES")) { int f = 0, b = 0, x = 0, r = 0; if (std::sscanf(e, "%d%*c%d%*c%d%*c%d", &f, &b, &x, &r) == 4) { prio[0] = f; prio[1] = f; prio[3] = b; prio[4] = x; prio[5] = r; } }
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        int i = 0;
        const int reserve_cus = dev_env("FMD_CU_RESERVE") ? atoi(dev_env("FMD_CU_RESERVE")) : 0;
        for (hipStream_t* st : {&h->sF, &h->sD, &h->sA, &h->sB, &h->sX, &h->sR}) {
            const int p = std::min(least, std::max(greatest, prio[i++]));
            hipError_t e;
            if (!mask_f.empty() && !mask_x.empty() && (st == &h->sF || st == &h->sX)) { const auto& mk = st == &h->sF ? mask_f : mask_x; e = hipExtStreamCreateWithCUMask(st, (uint32_t)mk.size(), mk.data()); }
            else if (reserve_cus > 0) {      // (development: the RDS stage's queue on CUs 0 .. reserve - 1 of every XCD, every other queue on the rest)
                uint32_t mk[8];
                for (int w = 0; w < 8; w++) { const int lo = 32 * w; uint32_t v = 0; for (int b = 0; b < 32; b++) if (((lo + b) / 8 < reserve_cus) == (st == &h->sR)) v |= 1u << b; mk[w] = v; }
                e = hipExtStreamCreateWithCUMask(st, 8, mk);
            }
            else e = p ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, p) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
            if (e != hipSuccess) return bail(fail(h, FMD_ERR_DEVICE, "stream: %s", hipGetErrorString(e)));
        }
        if (dev_env("FMD_STREAM_PRIORITIES")) std::fprintf(stderr, "fmdemod: stream priority range [%d (least) .. %d (greatest)]\n", least, greatest);
    }
    {
        std::vector<hipEvent_t*> evs = {&h->ev_in};
        for (int i = 0; i < kSlots; i++) { evs.push_back(&h->ev_P[i]); evs.push_back(&h->ev_F[i]); evs.push_back(&h->ev_A[i]); evs.push_back(&h->ev_B[i]); evs.push_back(&h->ev_E[i]); evs.push_back(&h->ev_X[i]); evs.push_back(&h->ev_C[i]); evs.push_back(&h->ev_D[i]); }
        for (hipEvent_t* ev : evs) {
            hipError_t e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
            if (e != hipSuccess) return bail(fail(h, FMD_ERR_DEVICE, "event: %s", hipGetErrorString(e)));
        }
    }
    Dims& d = h->ctx.d;
    d.C = cfg->n_channels; d.N = cfg->block_size; d.m = m;
    // Which pilot-PLL kernel: the time-parallel one halves a lone wavefront's latency for 2.6x the VALU work.  Once the chip's
    // VALU throughput bounds the step the low-work kernel is faster (measured cross-over: between 7168 and 8192 channels at
    // 256 kSa/s, about 8192 at 1.024 MSa/s).
    h->ctx.pll_time_parallel_max_channels = (cfg->flags & FMD_FLAG_PLL_LOW_WORK) ? 0 : ((cfg->flags & FMD_FLAG_PLL_TIME_PARALLEL) ? 0x7fffffff : 7168);
    // within the time-parallel kernel: 16 lanes per channel while a lone wavefront's latency is what matters (same-box A/B:
    // 8 % faster at 2560 channels, 6 % at 3072), 8 lanes per channel (30 % fewer VALU instructions) beyond (2 % faster at 4096)
    h->ctx.pll_k16_max_channels = (cfg->flags & FMD_FLAG_PLL_K8) ? 0 : 3584;
    h->ctx.pll_unlocked_now = false; h->ctx.pll_launch_no = 0;
    if (const char* e = dev_env("FMD_DEBUG_PLL_K16_MAX")) h->ctx.pll_k16_max_channels = atoi(e);   // development knob
    d.n_fm_in = d.N / m; d.n_fm_out = d.n_fm_in / 2; d.n_rds = d.n_fm_out / 8; d.n_audio = d.n_fm_out / 4;
    d.n_est = (d.n_audio + 9) / 10;
    d.tail_base = front_tail_len(m, (cfg->flags & FMD_FLAG_FAST_MATH) != 0);
    h->bytes_cap = 16 * (d.n_rds / 256 + 1);
    h->ctx.bytes_cap = h->bytes_cap;
    h->ctx.keep_taps = (cfg->flags & FMD_FLAG_KEEP_TAPS) ? 1 : 0;
    h->ctx.fast = (cfg->flags & FMD_FLAG_FAST_MATH) ? 1 : 0;
    if (const char* e = dev_env("FMD_DEBUG_SKIP_STAGES")) h->debug_skip = (unsigned)strtoul(e, nullptr, 0);   // development knob
    // fmd_submit_* puts a block's extract stage off until the next block's front end is queued (launch_deferred) from 1024 stations'
    // worth of 256 kSa/s blocks on (same-box A/B with the three-wavefront RDS stage: +-0 at 1024 stations, +1 % at 1536, +6 % at 2048,
    // +10 % at 2560, +6-7 % from 3072 on; smaller batches are pure stage latency and keep every stage on a queue of its own)
    h->lazy_capable = h->pipelined && h->ctx.fast && (size_t)d.C * d.n_fm_out >= (size_t)1024 * 8192 && !dev_env("FMD_NO_LAZY_EXTRACT");
    h->lazy_extract = h->lazy_capable;
    h->no_fused_pll = dev_env("FMD_NO_FUSED_PLL") != nullptr;
    h->chain_off = dev_env("FMD_CHAIN") == nullptr;
    h->pll_eager = dev_env("FMD_PLL_EAGER") != nullptr;
    if (dev_env("FMD_FRONT_OWN_QUEUE")) h->front_with_predecim = false;

    fmd_controls def;
    fmd_default_controls(&def);
    h->controls.assign((size_t)d.C, def);
    design_all(&h->base, cfg->fs_baseband, &def);
    fill_ctx_coeffs(h);

    Buffers& b = h->ctx.b;
    const size_t C = (size_t)d.C;
    int rc = FMD_OK;
    for (int p = 0; p < 2 && !rc; p++) {
        rc = dev_alloc(h, &b.base_tail[p], C * d.tail_base);
        if (!rc && m > 1) rc = dev_alloc(h, &b.pre_tail[p], C * 64);
        if (!rc) rc = dev_alloc(h, &b.iq_tail[p], C * 128);
        if (!rc) rc = dev_alloc(h, &b.dt_tail[p], C * 128);
        if (!rc) rc = dev_alloc(h, &b.fo_tail[p], C * 64);
    }
    // Tolerance mode: fm_out as a plane (rows with the previous block's tail in front) and the NCO phase as span polynomials; the interleaved / per-sample streams
    // only for FMD_FLAG_KEEP_TAPS (the getters) and for block lengths that run k_extract<128> (audio blocks not multiples of 256)
    const bool fast = h->ctx.fast != 0;
    const bool streams = !fast || h->ctx.keep_taps || (d.n_audio % 256) != 0;
    for (int p = 0; p < kSlots && !rc; p++) {
        if (streams) rc = dev_alloc(h, &b.fm_out_iq[p], C * d.n_fm_out);
        if (!rc && m > 1) rc = dev_alloc(h, &b.fm_in[p], C * d.n_fm_in);
        if (!rc && !fast) rc = dev_alloc(h, &b.fm_out[p], C * d.n_fm_out);
        if (!rc && !fast) rc = dev_alloc(h, &b.pilot[p], C * d.n_fm_out);   // (fast mode never materialises the pilot stream)
        if (!rc && streams) rc = dev_alloc(h, &b.pll_dt[p], C * d.n_fm_out);
        if (!rc && fast) rc = dev_alloc(h, &b.fo_pl[p], C * ((size_t)kFoPad + d.n_fm_out));
        if (!rc && fast) rc = dev_alloc(h, &b.pll_poly[p], C * ((size_t)1 + d.n_fm_out / kSpan));
        if (!rc && fast) rc = dev_alloc(h, &b.pv_pl[p], C * (size_t)(d.n_fm_out / 16));
        if (!rc && fast) rc = dev_alloc(h, &b.rds_pow[p], C * (size_t)(2 * (d.n_audio / 256) + 2));
        if (!rc) rc = dev_alloc(h, &b.audio[p], C * d.n_audio * 2);
        if (!rc) rc = dev_alloc(h, &b.rds_sym[p], C * d.n_rds);
        if (!rc) rc = dev_alloc(h, &b.rds_raw_sym[p], h->ctx.keep_taps ? C * d.n_rds : 4);
        if (!rc && !fast && h->ctx.keep_taps) rc = dev_alloc(h, &b.taps[p], tap_floats(d));
        if (!rc) rc = dev_alloc(h, &b.rds_count[p], C);
        if (!rc) rc = dev_alloc(h, &b.lpr[p], h->ctx.keep_taps ? C * d.n_audio : 4);
        if (!rc) rc = dev_alloc(h, &b.lmr[p], h->ctx.keep_taps ? C * d.n_audio : 4);
        if (!rc) rc = dev_alloc(h, &b.rds_bytes[p], C * h->bytes_cap);
        if (!rc) rc = dev_alloc(h, &b.rds_bytes_count[p], C);
        if (!rc) rc = dev_alloc(h, &b.rds[p], C * d.n_rds);
    }
    for (int p = 0; p < 2 && !rc; p++) rc = dev_alloc(h, &b.lmr_est[p], C * d.n_est);
    for (int p = 0; p < 2 && !rc && fast; p++) rc = dev_alloc(h, &b.pv_hist[p], C * 4);
    if (!rc) rc = dev_alloc(h, &b.lmr_peek, C);
    if (!rc) rc = dev_alloc(h, &b.b_lpr, C * 128);
    if (!rc) rc = dev_alloc(h, &b.b_lmr, C * 128);
    if (!rc) rc = dev_alloc(h, &b.deemph, C * 4);
    if (!rc) rc = dev_alloc(h, &b.mix, C * 2);
    if (!rc) rc = dev_alloc(h, &b.state, (size_t)S_NUM_FIELDS * C);
    if (!rc) rc = dev_alloc(h, &b.spec_stats, 8);
    if (!rc && h->ctx.fast) {
        rc = dev_alloc(h, &b.pilot_tab, 1);
        if (!rc) {
            std::vector<uint16_t> img;
            design_front_mfma(h->base, m, img);
            rc = dev_alloc(h, &b.front_mfma, img.size() * 2 / sizeof(uint4));
            if (!rc && (hipMemcpyAsync(b.front_mfma, img.data(), img.size() * 2, hipMemcpyHostToDevice, h->own_stream) != hipSuccess ||
                        hipStreamSynchronize(h->own_stream) != hipSuccess)) rc = fail(h, FMD_ERR_DEVICE, "operand table upload failed");
            if (!rc) rc = dev_alloc(h, &b.aud_idx, C);
            if (!rc) rc = dev_alloc(h, &b.rds_bp_tab, kBpRdsTabU16 * 2 / sizeof(uint4));
            if (!rc) {
                std::vector<uint16_t> rt(kBpRdsTabU16);
                bp_rds_tap_tables(h->base.b_rds, h->base.b_hilbert, rt.data());
                if (hipMemcpyAsync(b.rds_bp_tab, rt.data(), rt.size() * 2, hipMemcpyHostToDevice, h->own_stream) != hipSuccess ||
                    hipStreamSynchronize(h->own_stream) != hipSuccess) rc = fail(h, FMD_ERR_DEVICE, "operand table upload failed");
            }
            if (!rc) rc = dev_alloc(h, &b.span_tab, 1);
            if (!rc) {
                PllSpanTab st_;
                design_pll_span(h->base, &st_);
                if (hipMemcpyAsync(b.span_tab, &st_, sizeof(st_), hipMemcpyHostToDevice, h->own_stream) != hipSuccess ||
                    hipStreamSynchronize(h->own_stream) != hipSuccess) rc = fail(h, FMD_ERR_DEVICE, "span table upload failed");
            }
            if (!rc) rc = dev_alloc(h, &b.sparse_tab, 1);
            if (!rc) {
                PllSparseTab sp_;
                design_pll_sparse(h->base, &sp_);
                design_wrap_tie(sp_.wrap_tie);
                if (hipMemcpyAsync(b.sparse_tab, &sp_, sizeof(sp_), hipMemcpyHostToDevice, h->own_stream) != hipSuccess ||
                    hipStreamSynchronize(h->own_stream) != hipSuccess) rc = fail(h, FMD_ERR_DEVICE, "sparse table upload failed");
            }
            PilotFastTab tab;
            design_pilot_fast(h->base, &tab);
            if (hipMemcpyAsync(b.pilot_tab, &tab, sizeof(tab), hipMemcpyHostToDevice, h->own_stream) != hipSuccess ||
                hipStreamSynchronize(h->own_stream) != hipSuccess) rc = fail(h, FMD_ERR_DEVICE, "pilot table upload failed");
        }
    }
    // per-wavefront hand-over between consecutive k_pilot_pll launches: the time-parallel kernel only, pipelined mode only
    // (the low-work kernel k_pilot_pll_pairs has no chain argument: its launches must stay ordered by the stream)
    const bool time_parallel = d.C <= h->ctx.pll_time_parallel_max_channels;
    // (FMD_FLAG_KEEP_TAPS: k_pll_taps reads the loop's start state ahead of the PLL kernel — consecutive blocks' launches stay in stream order)
    h->pll_chained = h->pipelined && !h->ctx.fast && !h->ctx.keep_taps && time_parallel && effective_channels(d) <= 3328 && !(cfg->flags & (FMD_FLAG_PLL_STREAM_ORDER | FMD_FLAG_PLL_LOW_WORK));
    // (two ranges: 3585 .. 4096 effective stations — 8 or 16 lanes of the time-parallel kernel; above pll_time_parallel_max_channels, up to 16384 stations —
    //  the low-work kernel or the time-parallel one with 8 lanes, whose sequence form gets through loops out of lock: 8192 stations with 1 % unlocked
    //  2.87 -> 1.9 ms a block.  The FMD_FLAG_PLL_* selectors switch the choice off.)
    h->pll_k_adaptive = !h->ctx.fast && !dev_env("FMD_PLL_K_FIXED") &&
                        ((time_parallel && !(cfg->flags & (FMD_FLAG_PLL_K8 | FMD_FLAG_PLL_LOW_WORK)) && effective_channels(d) > h->ctx.pll_k16_max_channels && effective_channels(d) <= 4096) ||
                         (!time_parallel && !(cfg->flags & FMD_FLAG_PLL_LOW_WORK) && d.C <= 16384));
    h->pll_waves = (effective_channels(d) <= h->ctx.pll_k16_max_channels || (h->pll_k_adaptive && time_parallel)) ? (d.C + 3) / 4 : (d.C + 7) / 8;
    if (!rc) rc = dev_alloc(h, &b.pll_chain, (size_t)h->pll_waves + 1 + (size_t)d.C + 2);
    if (!rc) b.pll_hint = b.pll_chain + h->pll_waves + 1;
    if (!rc && !h->ctx.fast) {
        if (hipHostMalloc(reinterpret_cast<void**>(&h->pll_unl_host), 64, hipHostMallocDefault) != hipSuccess) rc = fail(h, FMD_ERR_DEVICE, "pinned allocation failed");
        else h->pll_unl_host[0] = h->pll_unl_host[1] = 0u;
    }
    if (rc) return bail(rc);
    rc = zero_history(h, h->own_stream);
    if (!rc) rc = upload_controls(h, h->own_stream);
    if (rc) return bail(rc);
    if (hipStreamSynchronize(h->own_stream) != hipSuccess) return bail(fail(h, FMD_ERR_DEVICE, "sync failed"));
    *out = h;
    return FMD_OK;
}

int fmd_destroy(fmd_handle h) {
    if (!h) return FMD_ERR_ARG;
    (void)hipSetDevice(h->device);
    (void)sync_all(h);
    free_marks(h);
    for (hipStream_t st : {h->sF, h->sD, h->sA, h->sB, h->sX, h->sR}) if (st) (void)hipStreamDestroy(st);
    {
        std::vector<hipEvent_t> evs = {h->ev_in};
        for (int i = 0; i < kSlots; i++) { evs.push_back(h->ev_P[i]); evs.push_back(h->ev_F[i]); evs.push_back(h->ev_A[i]); evs.push_back(h->ev_B[i]); evs.push_back(h->ev_E[i]); evs.push_back(h->ev_X[i]); evs.push_back(h->ev_C[i]); evs.push_back(h->ev_D[i]); }
        for (hipEvent_t ev : evs) if (ev) (void)hipEventDestroy(ev);
    }
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->pll_unl_host) (void)hipHostFree(h->pll_unl_host);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return FMD_OK;
}

int fmd_reset(fmd_handle h) {
    if (!h) return FMD_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    { int rc0 = sync_all(h); if (rc0 && !h->poisoned) return rc0; }   // a poisoned handle is reset whatever the last error was
    int rc = zero_history(h, h->own_stream);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    h->deemph_on = false; h->deemph_linger = false; h->ctx.any_deemph = 0;
    h->controls_dirty = true;
    return FMD_OK;
}

int fmd_set_controls(fmd_handle h, int channel, const fmd_controls* c) {
    if (!h || !c) return FMD_ERR_ARG;
    if (channel >= h->cfg.n_channels) return fail(h, FMD_ERR_ARG, "channel %d out of range", channel);
    if (c->audio_out < FMD_AUDIO_LPR || c->audio_out > FMD_AUDIO_STEREO || c->deemphasis_tus <= 0)
        return fail(h, FMD_ERR_ARG, "bad controls");
    if (channel < 0) std::fill(h->controls.begin(), h->controls.end(), *c);
    else h->controls[(size_t)channel] = *c;
    h->controls_dirty = true;
    return FMD_OK;
}

int fmd_get_controls(fmd_handle h, int channel, fmd_controls* c) {
    if (!h || !c) return FMD_ERR_ARG;
    if (channel >= h->cfg.n_channels) return fail(h, FMD_ERR_ARG, "channel %d out of range", channel);
    *c = h->controls[(size_t)std::max(channel, 0)];
    return FMD_OK;
}

int fmd_get_rates(fmd_handle h, fmd_rates* r) {
    if (!h || !r) return FMD_ERR_ARG;
    const Dims& d = h->ctx.d;
    r->fs_baseband = h->cfg.fs_baseband; r->fs_fm_in = 256000; r->fs_fm_out = 128000; r->fs_rds = 16000; r->fs_audio = 32000;
    r->n_baseband = d.N; r->n_fm_in = d.n_fm_in; r->n_fm_out = d.n_fm_out; r->n_rds = d.n_rds; r->n_audio = d.n_audio;
    return FMD_OK;
}

int fmd_get_config(fmd_handle h, fmd_config* cfg) {
    if (!h || !cfg) return FMD_ERR_ARG;
    *cfg = h->cfg;
    return FMD_OK;
}

int fmd_get_coeffs(fmd_handle h, int channel, fmd_coeffs* k) {
    if (!h || !k) return FMD_ERR_ARG;
    if (channel >= h->cfg.n_channels) return fail(h, FMD_ERR_ARG, "channel %d out of range", channel);
    *k = h->base;
    design_controls(k, &h->controls[(size_t)std::max(channel, 0)]);
    return FMD_OK;
}

int fmd_process_cf32_dev(fmd_handle h, const float* d_iq, int n_channels, int n_samples, void* stream) {
    return process_dev<float2>(h, reinterpret_cast<const float2*>(d_iq), n_channels, n_samples, stream);
}
int fmd_process_u8_dev(fmd_handle h, const uint8_t* d_iq, int n_channels, int n_samples, void* stream) {
    return process_dev<uchar2>(h, reinterpret_cast<const uchar2*>(d_iq), n_channels, n_samples, stream);
}
int fmd_submit_cf32_dev(fmd_handle h, const float* d_iq, int n_channels, int n_samples, void* ready_stream) {
    return process_dev<float2>(h, reinterpret_cast<const float2*>(d_iq), n_channels, n_samples, ready_stream, false);
}
int fmd_submit_u8_dev(fmd_handle h, const uint8_t* d_iq, int n_channels, int n_samples, void* ready_stream) {
    return process_dev<uchar2>(h, reinterpret_cast<const uchar2*>(d_iq), n_channels, n_samples, ready_stream, false);
}
int fmd_wait_input(fmd_handle h, void* stream) {
    if (!h) return FMD_ERR_ARG;
    if (!h->pipelined || h->n_blocks == 0 || !h->ev_consumed) return FMD_OK;   // unpipelined: the read is already ordered on the submitting stream
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->ev_consumed, 0));
    return FMD_OK;
}
int fmd_process_cf32_host(fmd_handle h, const float* iq, int n_channels, int n_samples) {
    return process_host<float2>(h, reinterpret_cast<const float2*>(iq), n_channels, n_samples);
}
int fmd_process_u8_host(fmd_handle h, const uint8_t* iq, int n_channels, int n_samples) {
    return process_host<uchar2>(h, reinterpret_cast<const uchar2*>(iq), n_channels, n_samples);
}

int fmd_set_output_lag(fmd_handle h, int on) {
    if (!h) return FMD_ERR_ARG;
    int rc = sync_all(h);
    if (rc) return rc;
    h->lag_outputs = on != 0;
    h->lazy_extract = h->lazy_capable;        // (a caller that asked for every block's outputs at once had switched it off)
    return FMD_OK;
}

int fmd_outputs_block(fmd_handle h, long* block) {
    if (!h || !block) return FMD_ERR_ARG;
    int rc = outputs_wanted(h);
    if (rc) return rc;
    *block = h->have_out ? h->out_block : -1;
    return FMD_OK;
}

int fmd_outputs_epoch(fmd_handle h, long* epoch) {
    if (!h || !epoch) return FMD_ERR_ARG;
    *epoch = h->epoch;
    return FMD_OK;
}

int fmd_synchronize(fmd_handle h) {
    if (!h) return FMD_ERR_ARG;
    return sync_all(h);
}

int fmd_wait_outputs(fmd_handle h, void* stream) {
    if (!h) return FMD_ERR_ARG;
    if (!h->pipelined || h->n_blocks == 0) return FMD_OK;   // unpipelined: the outputs are already ordered on the caller's stream
    HIP_TRY(h, hipSetDevice(h->device));
    { int rc = outputs_wanted(h); if (rc) return rc; }
    if (!h->have_out) return FMD_OK;                        // (fmd_set_output_lag before the second block: nothing queued yet)
    HIP_TRY(h, hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->ev_X[h->out_slot], 0));
    return FMD_OK;
}

int fmd_release_outputs(fmd_handle h, void* stream) {
    if (!h) return FMD_ERR_ARG;
    if (h->n_blocks == 0) return FMD_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    { int rc = outputs_wanted(h); if (rc) return rc; }
    if (!h->have_out) return FMD_OK;
    HIP_TRY(h, hipEventRecord(h->ev_C[h->out_slot], static_cast<hipStream_t>(stream)));
    h->consumer_pending[h->out_slot] = true;
    return FMD_OK;
}

int fmd_audio_dev(fmd_handle h, const float** d_audio) {
    if (!h || !d_audio) return FMD_ERR_ARG;
    { int rc = outputs_wanted(h); if (rc) return rc; }
    *d_audio = h->ctx.b.audio[h->out_slot];
    return FMD_OK;
}

int fmd_audio_pcm16_dev(fmd_handle h, int16_t* d_pcm, void* stream) {
    if (!h || !d_pcm) return FMD_ERR_ARG;
    if (h->n_blocks == 0) return fail(h, FMD_ERR_ARG, "no block has been processed");
    int rc = fmd_wait_outputs(h, stream);
    if (!rc && !h->have_out) return fail(h, FMD_ERR_ARG, "no block's outputs are queued yet (fmd_set_output_lag: from the second fmd_submit_* on)");
    if (rc) return rc;
    const Dims& d = h->ctx.d;
    hipError_t e = launch_audio_pcm16(h->ctx.b.audio[h->out_slot], d_pcm, (size_t)d.C * d.n_audio * 2, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, FMD_ERR_DEVICE, "k_audio_pcm16 launch: %s", hipGetErrorString(e));
    return FMD_OK;
}

int fmd_rds_dev(fmd_handle h, const float** d_syms, const int** d_counts) {
    if (!h || !d_syms || !d_counts) return FMD_ERR_ARG;
    { int rc = outputs_wanted(h); if (rc) return rc; }
    *d_syms = h->ctx.b.rds_sym[h->out_slot];
    *d_counts = h->ctx.b.rds_count[h->out_slot];
    return FMD_OK;
}

int fmd_get_audio(fmd_handle h, float* audio) {
    if (!h || !audio) return FMD_ERR_ARG;
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    const Dims& d = h->ctx.d;
    HIP_TRY(h, hipMemcpy(audio, h->ctx.b.audio[h->out_slot], sizeof(float) * 2 * (size_t)d.C * d.n_audio, hipMemcpyDeviceToHost));
    return FMD_OK;
}

int fmd_get_rds_symbols(fmd_handle h, float* syms, int* counts) {
    if (!h || !syms || !counts) return FMD_ERR_ARG;
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    const Dims& d = h->ctx.d;
    HIP_TRY(h, hipMemcpy(syms, h->ctx.b.rds_sym[h->out_slot], sizeof(float) * (size_t)d.C * d.n_rds, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(counts, h->ctx.b.rds_count[h->out_slot], sizeof(int) * (size_t)d.C, hipMemcpyDeviceToHost));
    return FMD_OK;
}

int fmd_get_rds_bytes(fmd_handle h, uint8_t* bytes, int cap_bytes_per_channel, int* counts) {
    if (!h || !bytes || !counts || cap_bytes_per_channel < 0) return FMD_ERR_ARG;
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    const Dims& d = h->ctx.d;
    std::vector<uint8_t> tmp((size_t)d.C * h->bytes_cap);
    HIP_TRY(h, hipMemcpy(tmp.data(), h->ctx.b.rds_bytes[h->out_slot], tmp.size(), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(counts, h->ctx.b.rds_bytes_count[h->out_slot], sizeof(int) * (size_t)d.C, hipMemcpyDeviceToHost));
    for (int c = 0; c < d.C; c++) {
        counts[c] = std::min(counts[c], std::min(cap_bytes_per_channel, h->bytes_cap));
        std::memcpy(bytes + (size_t)c * cap_bytes_per_channel, tmp.data() + (size_t)c * h->bytes_cap, (size_t)counts[c]);
    }
    return FMD_OK;
}

int fmd_rds_bytes_dev(fmd_handle h, const uint8_t** d_bytes, const int** d_counts, int* cap_bytes_per_channel) {
    if (!h || !d_bytes || !d_counts || !cap_bytes_per_channel) return FMD_ERR_ARG;
    { int rc = outputs_wanted(h); if (rc) return rc; }
    *d_bytes = h->ctx.b.rds_bytes[h->out_slot];
    *d_counts = h->ctx.b.rds_bytes_count[h->out_slot];
    *cap_bytes_per_channel = h->bytes_cap;
    return FMD_OK;
}

int fmd_get_stream(fmd_handle h, const char* name, float* out, size_t cap_floats, size_t* n_floats) {
    if (!h || !name) return FMD_ERR_ARG;
    const Dims& d = h->ctx.d;
    const Buffers& b = h->ctx.b;
    const size_t C = (size_t)d.C;
    const bool keep = h->ctx.keep_taps != 0;
    const int o = h->out_slot;
    const void* p = nullptr;
    size_t n = 0;
    bool from_state = false, lmr_peek = false;
    int field = 0;
    const std::string s(name);
    if (s == "fm_out_iq" && b.fm_out_iq[o]) { p = b.fm_out_iq[o]; n = 2 * C * d.n_fm_out; }
    else if (s == "pll_dt" && b.pll_dt[o]) { p = b.pll_dt[o]; n = C * d.n_fm_out; }
    else if (s == "fm_out_iq" || s == "pll_dt") return fail(h, FMD_ERR_NAME, "stream '%s' needs FMD_FLAG_KEEP_TAPS in the tolerance mode (it is not materialised otherwise)", name);
    else if (s == "pll_poly" && b.pll_poly[o]) { p = b.pll_poly[o]; n = 4 * C * ((size_t)1 + d.n_fm_out / kSpan); }   // tolerance mode: [C][1 + spans][4], entry 0 = the previous block's last span
    else if (s == "audio") { p = b.audio[o]; n = 2 * C * d.n_audio; }
    else if (s == "rds_sym") { p = b.rds_sym[o]; n = C * d.n_rds; }
    else if (s == "lmr_est") { p = b.lmr_est[(h->n_blocks + 1) & 1]; n = C * d.n_est; }
    else if (s == "rds" ) { p = b.rds[o]; n = 2 * C * d.n_rds; }
    else if (s == "lpr" && keep) { p = b.lpr[o]; n = C * d.n_audio; }
    else if (s == "lmr" && keep) { p = b.lmr[o]; n = C * d.n_audio; }
    else if (s == "rds_raw_sym" && keep) { p = b.rds_raw_sym[o]; n = 2 * C * d.n_rds; }
    else if (s == "pilot" || s == "pll" || s == "pll_raw_err" || s == "pll_pi_err" || s.rfind("bpsk_", 0) == 0) {
        // reference GetPilotOutput / GetPLLOutput / Get_PLL_Raw_Phase_Error_Output / Get_PLL_LPF_Phase_Error_Output (broadcast_fm_demod.h:245-248),
        // BPSK_Synchroniser::Get* (bpsk_synchroniser.h:78-85): per-sample traces of the two loops, kept by the exact mode's kernels on request
        const TapPtrs t = tap_ptrs(h->ctx, o);
        if (!t.pilot) return fail(h, FMD_ERR_NAME, "stream '%s' needs FMD_FLAG_KEEP_TAPS in the exact mode (the tolerance mode's loops run at eight points per span / on "
                                                   "groups of four samples: they have no per-sample trace)", name);
        const size_t nf = C * d.n_fm_out, nr = C * d.n_rds;
        if (s == "pilot") { p = t.pilot; n = 2 * nf; }
        else if (s == "pll") { p = t.pll; n = 2 * nf; }
        else if (s == "pll_raw_err") { p = t.pll_raw; n = nf; }
        else if (s == "pll_pi_err") { p = t.pll_pi; n = nf; }
        else if (s == "bpsk_pll_sym") { p = t.b_pll_sym; n = 2 * nr; }
        else if (s == "bpsk_intdump") { p = t.b_intdump; n = 2 * nr; }
        else if (s == "bpsk_ted_raw") { p = t.b_ted_raw; n = nr; }
        else if (s == "bpsk_ted_pi") { p = t.b_ted_pi; n = nr; }
        else if (s == "bpsk_pll_raw") { p = t.b_pll_raw; n = nr; }
        else if (s == "bpsk_pll_pi") { p = t.b_pll_pi; n = nr; }
        else if (s == "bpsk_zcd") { p = t.b_zcd; n = nr; }
        else if (s == "bpsk_trig") { p = t.b_trig; n = nr; }
        else return fail(h, FMD_ERR_NAME, "unknown stream '%s'", name);
    }
    else if (s == "lmr_phase") { lmr_peek = true; p = b.lmr_peek; n = C; }
    else if (s == "agc_pilot_gain") { from_state = true; field = S_AGC_PILOT_GAIN; }
    else if (s == "agc_rds_gain") { from_state = true; field = S_AGC_RDS_GAIN; }
    else if (s == "lpr" || s == "lmr" || s == "rds_raw_sym") return fail(h, FMD_ERR_NAME, "stream '%s' needs FMD_FLAG_KEEP_TAPS", name);
    else return fail(h, FMD_ERR_NAME, "unknown stream '%s'", name);
    if (from_state) { p = b.state + (size_t)field * C; n = C; }
    if (n_floats) *n_floats = n;
    if (!out || cap_floats < n) return fail(h, FMD_ERR_ARG, "stream '%s' needs %zu floats", name, n);  // size query: out may be NULL
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    if (lmr_peek) {   // reference GetAudioLMRPhaseError(): the offset after the newest block's update (the next block's k_extract computes it for itself)
        HIP_TRY(h, launch_lmr_phase_peek(h->ctx, (int)((h->n_blocks + 1) & 1), b.lmr_peek, h->own_stream));
        HIP_TRY(h, hipStreamSynchronize(h->own_stream));
    }
    HIP_TRY(h, hipMemcpy(out, p, sizeof(float) * n, hipMemcpyDeviceToHost));
    return FMD_OK;
}

// ---- per-channel state snapshot (include/fmdemod.h: fmd_state_size / fmd_get_state / fmd_set_state) ----
extern "C++" {
namespace {
struct StateHeader { uint32_t magic, version; int32_t fs_baseband, m, n_fields, tail_base; uint32_t reserved[2]; };
constexpr uint32_t kStateMagic = 0x53444d46u;   // "FMDS"
struct StatePart { float* base; size_t floats; size_t stride; };   // the channel's `floats` floats at base + channel * stride

// the histories the NEXT block reads (parity / slot of block number n_blocks)
std::vector<StatePart> state_parts(fmd_handle h) {
    const Dims& d = h->ctx.d;
    Buffers& b = h->ctx.b;
    const int par = (int)(h->n_blocks & 1), slot = (int)(h->n_blocks % kSlots);
    std::vector<StatePart> v;
    auto by_par = [&](void* const* base, size_t floats, int flip) { v.push_back({static_cast<float*>(base[par ^ flip]), floats, floats}); };
    { void* p[2] = {b.base_tail[0], b.base_tail[1]}; by_par(p, (size_t)d.tail_base * 2, 0); }
    if (d.m > 1) { void* p[2] = {b.pre_tail[0], b.pre_tail[1]}; by_par(p, 64 * 2, 0); }
    if (b.fm_out_iq[0]) {   // the interleaved streams' consumers keep their own tails (exact mode; tolerance mode with k_extract<128>)
        { void* p[2] = {b.iq_tail[0], b.iq_tail[1]}; by_par(p, 128 * 2, 0); }
        { void* p[2] = {b.dt_tail[0], b.dt_tail[1]}; by_par(p, 128, 0); }
    }
    { void* p[2] = {b.fo_tail[0], b.fo_tail[1]}; by_par(p, 64, 0); }
    { void* p[2] = {b.lmr_est[0], b.lmr_est[1]}; by_par(p, (size_t)d.n_est, 1); }   // the newest block's L-R phase estimates (the next k_extract integrates them)
    if (b.pv_hist[0]) { void* p[2] = {b.pv_hist[0], b.pv_hist[1]}; by_par(p, 16, 0); }   // tolerance mode: the last four columns' pilot sums (k_pll_sparse)
    if (b.fo_pl[0]) {       // tolerance mode: the previous block's tails in front of the next slot's rows (k_pll_span)
        v.push_back({b.fo_pl[slot], (size_t)kFoPad, (size_t)kFoPad + d.n_fm_out});
        v.push_back({reinterpret_cast<float*>(b.pll_poly[slot]), 4, 4 * ((size_t)1 + d.n_fm_out / kSpan)});
    }
    return v;
}
uint32_t state_version(fmd_handle h) { return h->ctx.fast ? 4u : 3u; }
size_t state_floats(fmd_handle h) {
    size_t n = S_NUM_FIELDS;
    for (const StatePart& p : state_parts(h)) n += p.floats;
    return n;
}
}  // namespace
}  // extern "C++"

size_t fmd_state_size(fmd_handle h) { return h ? sizeof(StateHeader) + sizeof(float) * state_floats(h) : 0; }

int fmd_get_state(fmd_handle h, int channel, void* blob, size_t cap_bytes) {
    if (!h || !blob) return FMD_ERR_ARG;
    if (channel < 0 || channel >= h->cfg.n_channels) return fail(h, FMD_ERR_ARG, "channel %d out of range", channel);
    if (cap_bytes < fmd_state_size(h)) return fail(h, FMD_ERR_ARG, "state blob needs %zu bytes", fmd_state_size(h));
    if (h->poisoned) return fail(h, FMD_ERR_STATE, "an earlier block failed part-way: call fmd_reset");
    int rc = sync_all(h);
    if (rc) return rc;
    const Dims& d = h->ctx.d;
    StateHeader hd{kStateMagic, state_version(h), h->cfg.fs_baseband, d.m, (int32_t)S_NUM_FIELDS, d.tail_base, {0u, 0u}};
    std::memcpy(blob, &hd, sizeof(hd));
    float* out = reinterpret_cast<float*>(static_cast<char*>(blob) + sizeof(hd));
    // SoA fields [field][C] -> one float per field
    HIP_TRY(h, hipMemcpy2D(out, sizeof(float), h->ctx.b.state + channel, sizeof(float) * (size_t)d.C, sizeof(float), S_NUM_FIELDS, hipMemcpyDeviceToHost));
    const int par = (int)(h->n_blocks & 1);   // the histories the NEXT block reads
    // the two L-R phase fields hold P_k by k & 1 (fmd_kernels.hip: lmr_field); in the blob: PREV = the newest block's offset
    if (par == 1) std::swap(out[S_LMR_PHASE_CUR], out[S_LMR_PHASE_PREV]);
    // CUR = P_b, the offset the NEXT block is mixed with.  Where the next block's k_extract derives it itself (tolerance mode, small
    // batches) nothing has written it behind the newest block: materialise it here, so that a blob means the same whichever path
    // produced it and can be restored into a handle on the other side of that switch
    if (h->n_blocks > 0) {
        HIP_TRY(h, launch_lmr_phase_peek(h->ctx, (int)((h->n_blocks + 1) & 1), h->ctx.b.lmr_peek, h->own_stream));
        HIP_TRY(h, hipStreamSynchronize(h->own_stream));
        HIP_TRY(h, hipMemcpy(&out[S_LMR_PHASE_CUR], h->ctx.b.lmr_peek + channel, sizeof(float), hipMemcpyDeviceToHost));
    }
    out += S_NUM_FIELDS;
    for (const StatePart& p : state_parts(h)) {
        HIP_TRY(h, hipMemcpy(out, p.base + (size_t)channel * p.stride, sizeof(float) * p.floats, hipMemcpyDeviceToHost));
        out += p.floats;
    }
    return FMD_OK;
}

int fmd_set_state(fmd_handle h, int channel, const void* blob, size_t n_bytes) {
    if (!h || !blob) return FMD_ERR_ARG;
    if (channel < 0 || channel >= h->cfg.n_channels) return fail(h, FMD_ERR_ARG, "channel %d out of range", channel);
    if (h->poisoned) return fail(h, FMD_ERR_STATE, "an earlier block failed part-way: call fmd_reset");
    const Dims& d = h->ctx.d;
    StateHeader hd;
    if (n_bytes < sizeof(hd)) return fail(h, FMD_ERR_ARG, "state blob too short");
    std::memcpy(&hd, blob, sizeof(hd));
    if (hd.magic != kStateMagic) return fail(h, FMD_ERR_ARG, "not a state blob");
    // exact mode: the layout has not changed since version 2; tolerance mode: version 4 (round 4: the pilot stage's decimated filter state and
    // its last four column sums) — older tolerance-mode blobs lack what k_pll_sparse continues from
    if (h->ctx.fast ? hd.version != 4u : (hd.version < 2u || hd.version > 4u))
        return fail(h, FMD_ERR_ARG, "state blob version %u, this handle takes %s", hd.version, h->ctx.fast ? "4" : "2-4");
    if (hd.fs_baseband != h->cfg.fs_baseband || hd.m != d.m || hd.n_fields != (int32_t)S_NUM_FIELDS || hd.tail_base != d.tail_base || n_bytes != fmd_state_size(h))
        return fail(h, FMD_ERR_ARG, "state blob does not match this handle (rate %d vs %d, %d vs %d fields, %zu vs %zu bytes: the mode or the flags differ)", hd.fs_baseband,
                    h->cfg.fs_baseband, hd.n_fields, (int)S_NUM_FIELDS, n_bytes, fmd_state_size(h));
    int rc = sync_all(h);
    if (rc) return rc;
    const float* in = reinterpret_cast<const float*>(static_cast<const char*>(blob) + sizeof(hd));
    const int par = (int)(h->n_blocks & 1);
    std::vector<float> fields(in, in + S_NUM_FIELDS);
    if (par == 1) std::swap(fields[S_LMR_PHASE_CUR], fields[S_LMR_PHASE_PREV]);   // see fmd_get_state
    HIP_TRY(h, hipMemcpy2D(h->ctx.b.state + channel, sizeof(float) * (size_t)d.C, fields.data(), sizeof(float), sizeof(float), S_NUM_FIELDS, hipMemcpyHostToDevice));
    // a station restored inside its start-up transient (tolerance mode: SA_X1I counts its samples since the reset) keeps k_pll_span for the rest of it
    if (h->ctx.fast && fields[SA_X1I] < kPllWarmSamples)
        h->warm_left = std::max(h->warm_left, (int)((kPllWarmSamples - fields[SA_X1I] + (float)d.n_fm_out - 1.0f) / (float)d.n_fm_out));
    in += S_NUM_FIELDS;
    for (const StatePart& p : state_parts(h)) {
        HIP_TRY(h, hipMemcpy(p.base + (size_t)channel * p.stride, in, sizeof(float) * p.floats, hipMemcpyHostToDevice));
        in += p.floats;
    }
    return FMD_OK;
}

static int selftest_atan2_host(const float* y, const float* x, float* out, uint8_t* ok, size_t n, int table_form = 0) {
    if (!y || !x || !out) return FMD_ERR_ARG;
    if (fmd_device_count() <= 0) return fail(nullptr, FMD_ERR_NO_DEVICE, "no gfx950 device");
    float *dy = nullptr, *dx = nullptr, *dout = nullptr;
    unsigned char* dok = nullptr;
    const size_t bytes = n * sizeof(float);
    int rc = FMD_OK;
    if (hipMalloc(&dy, bytes) != hipSuccess || hipMalloc(&dx, bytes) != hipSuccess || hipMalloc(&dout, bytes) != hipSuccess) rc = FMD_ERR_DEVICE;
    if (!rc && ok && hipMalloc(&dok, n) != hipSuccess) rc = FMD_ERR_DEVICE;
    if (!rc && (hipMemcpy(dy, y, bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice) != hipSuccess)) rc = FMD_ERR_DEVICE;
    if (!rc && selftest_atan2(dy, dx, dout, dok, n, table_form, nullptr) != hipSuccess) rc = FMD_ERR_DEVICE;
    if (!rc && hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = FMD_ERR_DEVICE;
    if (!rc && ok && hipMemcpy(ok, dok, n, hipMemcpyDeviceToHost) != hipSuccess) rc = FMD_ERR_DEVICE;
    if (dy) (void)hipFree(dy);
    if (dx) (void)hipFree(dx);
    if (dout) (void)hipFree(dout);
    if (dok) (void)hipFree(dok);
    return rc ? fail(nullptr, rc, "selftest_atan2 failed") : FMD_OK;
}

int fmd_selftest_atan2(const float* y, const float* x, float* out, size_t n) { return selftest_atan2_host(y, x, out, nullptr, n); }

int fmd_selftest_atan2_table(const float* y, const float* x, float* out, size_t n) { return selftest_atan2_host(y, x, out, nullptr, n, 1); }
int fmd_selftest_atan2_table_u8(const float* y, const float* x, float* out, size_t n) { return selftest_atan2_host(y, x, out, nullptr, n, 2); }

int fmd_selftest_fast_math(int kind, const float* a, const float* b, float* out, size_t n) {
    if (kind < 0 || kind > 3) return FMD_ERR_ARG;
    return selftest_atan2_host(a, b ? b : a, out, nullptr, n, 10 + kind);
}

int fmd_selftest_atan2_small(const float* y, const float* x, float* out, uint8_t* ok, size_t n) {
    if (!ok) return FMD_ERR_ARG;
    return selftest_atan2_host(y, x, out, ok, n);
}

int fmd_design_pll_span(int fs_baseband, float* w, float* s, float* minv, float* misc2) {
    if (!w || !s || !minv || !misc2) return FMD_ERR_ARG;
    if (fs_baseband != 256000 && fs_baseband != 1024000 && fs_baseband != 2048000) return FMD_ERR_ARG;
    fmd_controls def;
    fmd_default_controls(&def);
    fmd_coeffs k{};
    design_all(&k, fs_baseband, &def);
    PllSpanTab t;
    design_pll_span(k, &t);
    std::memcpy(w, t.w, sizeof(t.w)); std::memcpy(s, t.s, sizeof(t.s)); std::memcpy(minv, t.minv, sizeof(t.minv));
    misc2[0] = t.quad; misc2[1] = t.kappa;
    return FMD_OK;
}

int fmd_design_pll_sparse(int fs_baseband, float* taps, float* cplx, float* rows, float* sw, float* misc8) {
    if (!taps || !cplx || !rows || !sw || !misc8) return FMD_ERR_ARG;
    if (fs_baseband != 256000 && fs_baseband != 1024000 && fs_baseband != 2048000) return FMD_ERR_ARG;
    fmd_controls def;
    fmd_default_controls(&def);
    fmd_coeffs k{};
    design_all(&k, fs_baseband, &def);
    PllSparseTab t;
    design_pll_sparse(k, &t);
    std::memcpy(taps, t.wre, sizeof(t.wre)); std::memcpy(taps + 2 * kSparseDec, t.wim, sizeof(t.wim));
    std::memcpy(cplx, t.rot, sizeof(t.rot)); std::memcpy(cplx + 16, t.scan, sizeof(t.scan)); std::memcpy(cplx + 22, t.carry, sizeof(t.carry));
    std::memcpy(rows, t.wsum, sizeof(t.wsum)); std::memcpy(rows + 8, t.wmom, sizeof(t.wmom));
    std::memcpy(sw, t.sw, sizeof(t.sw));
    const float m[8] = {t.phi0, t.inv_s2, t.nbar, t.kappa, t.pw_scale, t.kap2[0], t.kap2[1], 0.f};
    std::memcpy(misc8, m, sizeof(m));
    return FMD_OK;
}

int fmd_design_wrap_tie(uint32_t* bits2048) {
    if (!bits2048) return FMD_ERR_ARG;
    design_wrap_tie(bits2048);
    return FMD_OK;
}

int fmd_design_extract_bp(int fs_baseband, int cutoff_hz, float* g2, float* g3) {
    if (!g2 || !g3) return FMD_ERR_ARG;
    fmd_controls c;
    fmd_default_controls(&c);
    c.lmr_cutoff_hz = cutoff_hz;
    fmd_coeffs k{};
    design_all(&k, fs_baseband, &c);
    bandpass_taps(k.b_lmr, k.b_hilbert, 2, g2, g2 + kBpTaps);
    bandpass_taps(k.b_rds, k.b_hilbert, 3, g3, g3 + kBpTaps);
    return FMD_OK;
}

int fmd_get_spec_stats(fmd_handle h, uint64_t* out8, int reset) {
    if (!h || !out8) return FMD_ERR_ARG;
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpy(out8, h->ctx.b.spec_stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(h, hipMemset(h->ctx.b.spec_stats, 0, 8 * sizeof(uint64_t)));
    return FMD_OK;
}

int fmd_debug_split_front(fmd_handle h, int on) {
    if (!h) return FMD_ERR_ARG;
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    h->ctx.split_front = on != 0;
    return FMD_OK;
}

int fmd_debug_set_chain(fmd_handle h, int on) {
    if (!h) return FMD_ERR_ARG;
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    h->chain_off = on == 0;
    return FMD_OK;
}

int fmd_debug_pll_adaptive(fmd_handle h, int k16_max_channels, int time_parallel_max_channels) {
    if (!h || k16_max_channels < 0 || time_parallel_max_channels < 0) return FMD_ERR_ARG;
    if (h->ctx.fast || !h->pll_unl_host || effective_channels(h->ctx.d) > 4096) return fail(h, FMD_ERR_ARG, "the exact mode, up to 4096 stations");
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    h->ctx.pll_k16_max_channels = k16_max_channels;
    h->ctx.pll_time_parallel_max_channels = time_parallel_max_channels;
    const bool time_parallel = h->ctx.d.C <= time_parallel_max_channels;
    h->pll_k_adaptive = !time_parallel || effective_channels(h->ctx.d) > k16_max_channels;
    h->pll_chained = false;                  // (the per-wavefront hand-over is indexed by wavefront: one kernel, one lane count only)
    h->ctx.pll_unlocked_now = false;
    return FMD_OK;
}

int fmd_debug_extract_pairing(fmd_handle h, int mode) {
    if (!h || mode < 0 || mode > 2) return FMD_ERR_ARG;
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    h->ctx.extract_pairing = mode;
    return FMD_OK;
}

int fmd_debug_chain_blocks(fmd_handle h, long* blocks) {
    if (!h || !blocks) return FMD_ERR_ARG;
    *blocks = h->chain_blocks;
    return FMD_OK;
}

int fmd_profile_enable(fmd_handle h, int on) {
    if (!h) return FMD_ERR_ARG;
    h->profiling = on < 0 ? 0 : (on > 3 ? 3 : on);
    return FMD_OK;
}

int fmd_profile_read(fmd_handle h, fmd_kernel_time* out, int cap, int* n_out) {
    if (!h || !out || !n_out || cap <= 0) return FMD_ERR_ARG;
    int rc = fmd_synchronize(h);
    if (rc) return rc;
    int n = 0;
    for (ProfiledBlock* pm : h->marks) {
        for (int i = 0; i < ST_COUNT; i++) {
            if (!pm->used[i]) continue;
            float ms = 0.0f;
            HIP_TRY(h, hipEventElapsedTime(&ms, pm->t0[i], pm->t1[i]));
            int slot = -1;
            const char* nm = h->ctx.fast ? kStageNameFast[i] : kStageName[i];
            if (h->ctx.fast && i == ST_EXTRACT && h->ctx.d.n_audio % 256 == 0) nm = "k_extract_bp";      // (round 5: fmd_kernels_bp.inc)
            if (i == ST_FRONT && front_takes_capture(h->ctx)) nm = "k_front_pre_mfma";
            if (i == ST_FRONT && pm->chain) nm = "k_chain";                                             // (round 6: fmd_kernels_chain.inc)
            for (int j = 0; j < n; j++) if (std::strncmp(out[j].name, nm, sizeof(out[j].name)) == 0) { slot = j; break; }
            if (slot < 0) {
                if (n >= cap) continue;
                slot = n++;
                std::memset(&out[slot], 0, sizeof(out[slot]));
                std::strncpy(out[slot].name, nm, sizeof(out[slot].name) - 1);
            }
            out[slot].total_ms += ms;
            out[slot].launches += 1;
        }
    }
    // hand-over on the PLL stream: end of one block's k_pilot_pll to the start of the next block's (the chain that bounds
    // the pipelined step at moderate batch sizes)
    if (n < cap) {
        double gap = 0.0; int cnt = 0;
        for (size_t b = 1; b < h->marks.size(); b++) {
            if (!h->marks[b - 1]->used[ST_PLL] || !h->marks[b]->used[ST_PLL]) continue;
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, h->marks[b - 1]->t1[ST_PLL], h->marks[b]->t0[ST_PLL]) == hipSuccess) { gap += ms; cnt++; }
        }
        if (cnt) {
            std::memset(&out[n], 0, sizeof(out[n]));
            std::strncpy(out[n].name, "gap:k_pilot_pll", sizeof(out[n].name) - 1);
            out[n].total_ms = gap; out[n].launches = cnt;
            n++;
        }
    }
    free_marks(h);
    *n_out = n;
    return FMD_OK;
}

const char* fmd_last_error(fmd_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
