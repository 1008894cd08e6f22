// Wideband channeliser (BASELINE configs[4]): one wideband IQ capture (e.g. 10 MSa/s) is split into C
// FM stations at the demodulator's 256 kSa/s, laid out [C][n_out] cf32 — exactly what fmd_process_cf32_dev consumes.
//
// NOT in the reference (it tunes one station in the RTL-SDR hardware, src/device): parity is unpinned; the tests validate
// by construction against a float64 restatement of the same definition (tests/test_channelizer.py).
//
// Definition.  fs_out / fs_in = L / M in lowest terms (256 k / 10 M = 16 / 625).  For station k with centre f_k:
//     x_k[n] = x[n] * exp(-j 2 pi f_k n / fs_in)                                 (mix to baseband; n absolute, 64-bit)
//     y_k[o] = sum_{t=0}^{T-1} h[p + L t] * x_k[n0 - t],   n0 = floor(o M / L),  p = (o M) mod L
// i.e. the polyphase form of "zero-stuff by L, low-pass h, keep every M-th": a polyphase filter bank whose branches are
// the L phases of one Kaiser-windowed prototype (cut-off fs_out/2, length L*T), shared by all stations.  Per output
// sample that is T complex-by-real MACs; 40 stations x 256 kSa/s x 640 taps = 13 GFMA/s — noise next to the demodulator.
//
// Kernel (round 2): workgroup = (station, 128 consecutive outputs), 256 threads, all of them on the FIR.
//   * The input window those outputs need (128 M/L + T samples) is mixed to baseband while it is staged into LDS.  The mixer's
//     phasor is computed exactly (64-bit modular phase, sincospi) for a thread's first sample and advanced by a complex
//     multiplication for its following ones (stride 256 samples, ~22 steps: drift < 2e-6), instead of one sincospi per sample.
//   * Outputs o and o + L share their polyphase branch p = (o M) mod L.  Thread (p, s) owns branch p and the tap slice
//     [s T/16, (s+1) T/16) of it — its taps live in registers — and accumulates that slice for the tile's 128 / L outputs of
//     the branch: one tap load per (128 / L) x 2 FMAs instead of one per 2.  The 16 slices of an output are summed through LDS
//     in a fixed order.
// Requires L == 16 (10 MSa/s -> 256 kSa/s and every other pair of rates with fs_out / gcd = 16) for this mapping; other ratios
// take the general kernel below (one output per thread), which is what round 1 had.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "fmdemod.h"
#include "fmd_kernels.h"     // dev_env: development switches are read only in builds with -DFMD_DEV_HOOKS

namespace {

constexpr int kTile = 128;          // outputs per workgroup
constexpr int kMaxWindow = 7168;    // staged input samples per workgroup (56 KB of LDS)
constexpr int kSlices = 16;         // tap slices per branch in the L == 16 kernel

struct ChanDims {
    int L, M, T;            // interpolation, decimation, taps per phase
    int n_stations;
    long long n_out;        // outputs per station this call
    long long out_stride;   // row stride of the caller's output buffer (its capacity per station)
    unsigned long long o0;  // absolute index of the first output of this call
    unsigned long long n_base;  // absolute input index of the window's first sample: [T - 1 history samples][this call's block]
    long long n_in;             // samples of this call's block
};

// The window of a call is [T - 1 history samples][the caller's block]: two buffers, read in place (round 5: staging the block behind the
// history and copying the new history out were two device copies around every launch — 5 us each plus the queue's gaps, a quarter of a
// 10 MSa/s block's 100 us on the channeliser's queue).  The launch's first workgroup hands the last T - 1 samples over to the OTHER
// history buffer, which no workgroup of this launch reads.
struct ChanWin { const float2* hist; const float2* blk; float2* next_hist; };
__device__ __forceinline__ float2 win_at(const ChanDims& d, const ChanWin& w, long long j) {
    // (base and index are selected, then ONE load, no branch; the address is never formed from `blk` with a negative offset — ADVICE r5)
    const long long jb = j - (d.T - 1);
    const float2* base = jb < 0 ? w.hist : w.blk;
    return base[jb < 0 ? j : jb];
}
__device__ __forceinline__ void hand_over_history(const ChanDims& d, const ChanWin& w) {
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < d.T - 1; i += 256) w.next_hist[i] = win_at(d, w, d.n_in + i);
}
// window staging shared by both kernels: xs[i] = win[n_lo + i] * exp(-j 2 pi f n / fs)
__device__ __forceinline__ void stage_mixed(const ChanDims& d, const ChanWin& win, unsigned long long n_lo, int n_win,
                                            unsigned long long inc, float2* xs) {
    // phase in turns = frac(n_abs * f_k / fs_in), exact in 64-bit modular arithmetic, then one rounding to float
    const unsigned long long n_first = n_lo + (unsigned long long)threadIdx.x;
    const unsigned int ph0 = (unsigned int)((n_first * inc) >> 32);
    float s, c;
    sincospif((float)ph0 * 4.656612873077393e-10f /* 2^-31: argument in units of pi */, &s, &c);
    const unsigned int phs = (unsigned int)((256ull * inc) >> 32);          // phase advance between a thread's consecutive samples
    float ss, cs;
    sincospif((float)phs * 4.656612873077393e-10f, &ss, &cs);
    for (int i = threadIdx.x; i < n_win; i += 256) {
        const float2 x = win_at(d, win, (long long)((n_lo + (unsigned long long)i) - d.n_base));
        xs[i] = make_float2(fmaf(x.x, c, x.y * s), fmaf(x.y, c, -(x.x * s)));   // x * (cos - j sin)
        const float cn = fmaf(c, cs, -(s * ss)), sn = fmaf(s, cs, c * ss);
        c = cn; s = sn;
    }
}

// L == 16: see the header comment
__global__ __launch_bounds__(256) void k_channelize16(ChanDims d, ChanWin win, const float* __restrict__ taps /* [T][16] */,
                                                      const unsigned long long* __restrict__ phase_inc, float2* __restrict__ out) {
    constexpr int L = 16, PER = kTile / L;                    // 8 outputs of a branch per tile
    constexpr int kMaxSlice = 64;                             // taps per slice held in registers (T <= 1024)
    __shared__ float2 xs[kMaxWindow];
    __shared__ float2 part[kSlices][kTile + 1];
    const int k = blockIdx.y;
    const long long tile0 = (long long)blockIdx.x * kTile;
    const int n_tile = (int)((d.n_out - tile0) < kTile ? (d.n_out - tile0) : kTile);
    const unsigned long long o_first = d.o0 + (unsigned long long)tile0, o_last = o_first + (unsigned long long)(n_tile - 1);
    const unsigned long long n_hi = (o_last * (unsigned long long)d.M) / (unsigned long long)L;
    const unsigned long long n_lo = (o_first * (unsigned long long)d.M) / (unsigned long long)L - (unsigned long long)(d.T - 1);
    stage_mixed(d, win, n_lo, (int)(n_hi - n_lo + 1), phase_inc[k], xs);
    hand_over_history(d, win);
    // this thread's branch and slice: outputs oo = j0 + 16 q (q < 8) where j0 is the tile-relative output with branch p
    const int p = threadIdx.x & (L - 1), sl = threadIdx.x >> 4;
    const int tps = d.T / kSlices, t0 = sl * tps;             // T is a multiple of 64
    // tile-relative output whose branch is p: (o_first + j) M mod 16 == p.  M mod 16 is odd for coprime L, M: solve by search
    int j0 = 0;
    for (int j = 0; j < L; j++) if ((int)(((o_first + (unsigned long long)j) * (unsigned long long)d.M) % L) == p) j0 = j;
    float h[kMaxSlice];
#pragma unroll
    for (int t = 0; t < kMaxSlice; t++) h[t] = (t < tps) ? taps[(size_t)(t0 + t) * L + p] : 0.0f;
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < PER; q++) {
        const int oo = j0 + L * q;
        float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
        if (oo < n_tile) {
            const unsigned long long om = (o_first + (unsigned long long)oo) * (unsigned long long)d.M;
            const int n0 = (int)(om / (unsigned long long)L - n_lo) - t0;
#pragma unroll
            for (int t = 0; t < kMaxSlice; t += 2) {
                if (t < tps) {
                    const float2 x0 = xs[n0 - t], x1 = xs[n0 - t - 1];
                    ar = fmaf(h[t], x0.x, ar); ai = fmaf(h[t], x0.y, ai);
                    br = fmaf(h[t + 1], x1.x, br); bi = fmaf(h[t + 1], x1.y, bi);
                }
            }
        }
        if (oo < kTile) part[sl][oo] = make_float2(ar + br, ai + bi);
    }
    __syncthreads();
    if (threadIdx.x < n_tile) {
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < kSlices; s2++) { const float2 v = part[s2][threadIdx.x]; sr += v.x; si += v.y; }
        out[(size_t)k * d.out_stride + tile0 + threadIdx.x] = make_float2(sr, si);
    }
}

// L == 16 on the matrix cores (round 4).  The 16 consecutive outputs o = 16 g + m of a "group" g use each polyphase branch once
// (p = m M mod 16) and read the input window x[g M - (T - 1) + tau], tau < K = floor(15 M / 16) + T (1225 at 625 / 640):
//     y[16 g + m] = sum_tau A[m][tau] x[g M - (T - 1) + tau],   A[m][tau] = h[p(m) + 16 t],  t = floor(m M / 16) + T - 1 - tau  (0 <= t < T, else 0)
// — one banded 16 x K operand for EVERY group of every station (52 % dense).  D = A X with X's columns = the windows of a tile's 8 groups,
// real rail in columns 0-7 and imaginary rail in 8-15: v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulation: the arithmetic of the
// VALU form, no bf16 split; a column's window starts at any sample, operand reads are single floats).  Wavefront w holds its quarter of
// A's K-steps in registers (77 floats per lane) for every tile the workgroup takes; the four partial tiles meet in LDS.
// The VALU form above read every staged sample from LDS once per tap (2 FMAs per 8-byte read, 2-4-way bank conflicts): here a read
// feeds 16 rows, and the rails' column strides (M = 625 floats, rails 5608 apart) put a wavefront's 32-lane passes on distinct banks.
constexpr int kMG = 8;                      // groups (of 16 outputs) per tile: kTile outputs
constexpr int kMKW = 77;                    // K-steps of 4 per wavefront: K <= 4 * 4 * 77 = 1232
constexpr int kMK = 16 * kMKW;              // 1232
constexpr int kMRail = 5608;                // floats per rail: (kMG - 1) M + kMK <= 5608, and 5608 mod 32 == 8 (bank offset between the rails)
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Non-finite input: the banded operand multiplies EVERY staged sample of a group's 1232-sample window, its structural zeros included, so
// one Inf / NaN input sample makes all 16 outputs of every group whose window holds it NaN (the VALU form only those with a real tap on
// it).  A capture from a device is u8 / finite by construction; a host that may feed non-finite floats filters them at its boundary.
__global__ __launch_bounds__(256, 3) void k_channelize16_mfma(ChanDims d, ChanWin win, const float* __restrict__ atab /* [4][77][64] */,
                                                           const unsigned long long* __restrict__ phase_inc, float2* __restrict__ out, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xr = smem;                       // mixed window, real rail: xr[i] = Re x_k[n_lo + i]
    float* xi = smem + kMRail;
    float* part = smem + 2 * kMRail;        // [wavefront][column][row]
    const int k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 15, kq = lane >> 4;
    float a[kMKW];
#pragma unroll
    for (int j = 0; j < kMKW; j++) a[j] = atab[(size_t)(wv * kMKW + j) * 64 + lane];
    const unsigned long long inc = phase_inc[k];
    // the phase advance between a thread's consecutive samples (stride 256): see stage_mixed
    float ss, cs;
    sincospif((float)(unsigned int)((256ull * inc) >> 32) * 4.656612873077393e-10f, &ss, &cs);
    const float* xb = ((col >> 3) ? xi : xr) + d.M * (col & 7) + kq + 4 * kMKW * wv;
    hand_over_history(d, win);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long tile0 = (long long)tile * kTile;
        const int n_tile = (int)((d.n_out - tile0) < kTile ? (d.n_out - tile0) : kTile);
        const unsigned long long o_first = d.o0 + (unsigned long long)tile0;                  // a multiple of 16
        const unsigned long long n_lo = (o_first * (unsigned long long)d.M) / 16ull - (unsigned long long)(d.T - 1);
        const int n_valid = (int)(((o_first + (unsigned long long)(n_tile - 1)) * (unsigned long long)d.M) / 16ull - n_lo) + 1;   // samples that exist
        {
            float s, c;
            sincospif((float)(unsigned int)(((n_lo + (unsigned long long)tid) * inc) >> 32) * 4.656612873077393e-10f, &s, &c);
            const long long j0 = (long long)(n_lo - d.n_base);
            constexpr int PER = (kMRail + 255) / 256;
            float2 x[PER];
            // the caller's block through ONE base and immediate offsets (per-sample pointers for all 22 spilled registers); the history's
            // samples — in front of a call's first windows only — are patched in behind, a sample at a time
            const long long jb = j0 - (d.T - 1);          // where the window starts in the block: negative in front of a call's first windows
            // (the base of the immediate offsets, made by INTEGER arithmetic: with jb < 0 it lies in front of the block and `win.blk + jb` would be
            //  pointer arithmetic out of the array — ADVICE r5; only elements i >= -jb are ever read through it)
            const float2* src = reinterpret_cast<const float2*>(reinterpret_cast<uintptr_t>(win.blk) + (uintptr_t)(jb * (long long)sizeof(float2)));
            const int i_lo = jb < 0 ? (int)(-jb) : 0;     // (one unsigned compare per sample: i_lo <= i < n_valid)
#pragma unroll
            for (int r = 0; r < PER; r++) { const int i = tid + 256 * r; x[r] = ((unsigned)(i - i_lo) < (unsigned)(n_valid - i_lo)) ? src[i] : make_float2(0.f, 0.f); }
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int i = tid + 256 * r;
                if (i < kMRail) { xr[i] = fmaf(x[r].x, c, x[r].y * s); xi[i] = fmaf(x[r].y, c, -(x[r].x * s)); }   // x * (cos - j sin)
                const float cn = fmaf(c, cs, -(s * ss)), sn = fmaf(s, cs, c * ss);
                c = cn; s = sn;
            }
            if (jb < 0) {
                const int nh = (int)(-jb) < n_valid ? (int)(-jb) : n_valid;
#pragma unroll 1
                for (int i = tid; i < nh; i += 256) {
                    const float2 v = win.hist[j0 + i];
                    // the phase of sample i exactly as the recurrence above reaches it: i = tid + 256 r steps of (cs, ss) from the thread's first
                    float s2, c2;
                    sincospif((float)(unsigned int)(((n_lo + (unsigned long long)tid) * inc) >> 32) * 4.656612873077393e-10f, &s2, &c2);
                    for (int q = 0; q < i / 256; q++) { const float cn = fmaf(c2, cs, -(s2 * ss)), sn = fmaf(s2, cs, c2 * ss); c2 = cn; s2 = sn; }
                    xr[i] = fmaf(v.x, c2, v.y * s2); xi[i] = fmaf(v.y, c2, -(v.x * s2));
                }
            }
        }
        __syncthreads();
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kMKW; j++) {
            if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], xb[4 * j], acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], xb[4 * j], acc0, 0, 0, 0);
        }
        acc0 = acc0 + acc1;
        *reinterpret_cast<float4*>(part + (wv * 16 + col) * 16 + 4 * kq) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
        __syncthreads();
        if (tid < n_tile) {
            const int g = tid >> 4, m = tid & 15;
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) { re += part[(w * 16 + g) * 16 + m]; im += part[(w * 16 + 8 + g) * 16 + m]; }
            out[(size_t)k * d.out_stride + tile0 + tid] = make_float2(re, im);
        }
    }
}

// any L <= 64: one output per thread (the first 128 threads), taps read from global memory
__global__ __launch_bounds__(256) void k_channelize(ChanDims d, ChanWin win, const float* __restrict__ taps /* [T][L] */,
                                                    const unsigned long long* __restrict__ phase_inc /* [C], turns * 2^64 per input sample */,
                                                    float2* __restrict__ out /* [C][out_stride] */) {
    __shared__ float2 xs[kMaxWindow];
    const int k = blockIdx.y;
    const long long tile0 = (long long)blockIdx.x * kTile;
    const int n_tile = (int)((d.n_out - tile0) < kTile ? (d.n_out - tile0) : kTile);
    // absolute input range needed by this tile: [n_lo, n_hi]
    const unsigned long long o_first = d.o0 + (unsigned long long)tile0, o_last = o_first + (unsigned long long)(n_tile - 1);
    const unsigned long long n_hi = (o_last * (unsigned long long)d.M) / (unsigned long long)d.L;
    const unsigned long long n_lo = (o_first * (unsigned long long)d.M) / (unsigned long long)d.L - (unsigned long long)(d.T - 1);
    stage_mixed(d, win, n_lo, (int)(n_hi - n_lo + 1), phase_inc[k], xs);
    hand_over_history(d, win);
    __syncthreads();
    for (int oo = threadIdx.x; oo < n_tile; oo += 256) {
        const unsigned long long o = o_first + (unsigned long long)oo;
        const unsigned long long om = o * (unsigned long long)d.M;
        const int n0 = (int)(om / (unsigned long long)d.L - n_lo);
        const int p = (int)(om % (unsigned long long)d.L);
        float ar[4] = {0.f, 0.f, 0.f, 0.f}, ai[4] = {0.f, 0.f, 0.f, 0.f};
        const float* tp = taps + p;
        for (int t = 0; t < d.T; t += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float h = tp[(size_t)(t + u) * d.L];
                const float2 x = xs[n0 - t - u];
                ar[u] = fmaf(h, x.x, ar[u]); ai[u] = fmaf(h, x.y, ai[u]);
            }
        }
        out[(size_t)k * d.out_stride + tile0 + oo] = make_float2((ar[0] + ar[2]) + (ar[1] + ar[3]), (ai[0] + ai[2]) + (ai[1] + ai[3]));
    }
}

double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 64; k++) { term *= (x / (2.0 * k)) * (x / (2.0 * k)); sum += term; if (term < 1e-18 * sum) break; }
    return sum;
}

thread_local std::string g_chan_error;

}  // namespace

struct fmd_channelizer_s {
    int device = 0;
    int L = 0, M = 0, T = 0, C = 0;
    double fs_in = 0, fs_out = 0;
    size_t max_in = 0;
    unsigned long long n_abs = 0;     // absolute index of the next input sample
    unsigned long long o_abs = 0;     // absolute index of the next output sample
    float2* win[2] = {nullptr, nullptr};   // [T - 1] history samples each, ping-pong: a launch reads one and writes the next call's into the other
    int cur = 0;                      // window the next call stages into (its first T-1 samples hold the history)
    hipEvent_t done = nullptr;        // end of the previous call's work, for callers that change streams between calls
    bool have_done = false;
    float* taps = nullptr;            // [T][L]
    float* atab = nullptr;            // k_channelize16_mfma's operand A, per wavefront, K-step and lane; null: that form does not apply
    unsigned long long* inc = nullptr;
    std::vector<float> h_taps;
    std::string err;
};

static int chan_fail(fmd_channelizer h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_chan_error = buf;
    return code;
}

extern "C" {

int fmd_chan_design(double fs_in, double fs_out, int taps_per_phase, float* taps /* [T][L], may be NULL */, int* L_out, int* M_out) {
    if (!(fs_in > 0) || !(fs_out > 0) || fs_out > fs_in || taps_per_phase <= 0 || taps_per_phase % 4 != 0) return FMD_ERR_ARG;
    const long long a = llround(fs_out), b = llround(fs_in);
    if (std::fabs(fs_out - (double)a) > 1e-9 || std::fabs(fs_in - (double)b) > 1e-9) return FMD_ERR_ARG;   // integer rates only
    const long long g = std::gcd(a, b);
    const long long L = a / g, M = b / g;
    if (L > 64 || M > (1 << 20)) return FMD_ERR_ARG;
    if (L_out) *L_out = (int)L;
    if (M_out) *M_out = (int)M;
    if (!taps) return FMD_OK;
    // Kaiser-windowed sinc at the up-sampled rate L*fs_in, cut-off fs_out/2, DC gain L (makes up for the zero-stuffing)
    const int T = taps_per_phase, N = (int)L * T;
    const double fc = 0.5 / (double)M;           // cut-off (fs_out / 2) as a fraction of the up-sampled rate: (fs_out/2) / (L fs_in)
    const double beta = 5.65326;                 // 60 dB
    const double mid = 0.5 * (N - 1), i0b = bessel_i0(beta);
    std::vector<double> h(N);
    double sum = 0.0;
    for (int n = 0; n < N; n++) {
        const double x = (double)n - mid;
        const double s = (x == 0.0) ? 2.0 * fc : std::sin(2.0 * M_PI * fc * x) / (M_PI * x);
        const double r = x / mid;
        h[n] = s * bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
        sum += h[n];
    }
    for (int n = 0; n < N; n++) {
        const int p = n % (int)L, t = n / (int)L;
        taps[(size_t)t * L + p] = (float)(h[n] * (double)L / sum);
    }
    return FMD_OK;
}

int fmd_chan_create(const fmd_chan_config* cfg, fmd_channelizer* out) {
    if (!cfg || !out || cfg->n_stations <= 0 || !cfg->center_hz || cfg->max_input_samples <= 0) return chan_fail(nullptr, FMD_ERR_ARG, "bad channeliser configuration");
    if (fmd_device_count() <= 0) return chan_fail(nullptr, FMD_ERR_NO_DEVICE, "no gfx950 device");
    int dev = cfg->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return chan_fail(nullptr, FMD_ERR_DEVICE, "hipGetDevice failed");
    const int T = cfg->taps_per_phase > 0 ? cfg->taps_per_phase : 640;
    int L = 0, M = 0;
    if (fmd_chan_design(cfg->fs_in, cfg->fs_out, T, nullptr, &L, &M) != FMD_OK) return chan_fail(nullptr, FMD_ERR_ARG, "unsupported rates %g -> %g (need integer rates, L <= 64)", cfg->fs_in, cfg->fs_out);
    if ((long long)kTile * M / L + T + 2 > kMaxWindow) return chan_fail(nullptr, FMD_ERR_ARG, "decimation %d/%d with %d taps per phase needs a larger staging window", M, L, T);
    fmd_channelizer h = new fmd_channelizer_s();
    h->device = dev; h->L = L; h->M = M; h->T = T; h->C = cfg->n_stations; h->fs_in = cfg->fs_in; h->fs_out = cfg->fs_out;
    h->max_in = (size_t)cfg->max_input_samples;
    h->h_taps.resize((size_t)T * L);
    fmd_chan_design(cfg->fs_in, cfg->fs_out, T, h->h_taps.data(), nullptr, nullptr);
    std::vector<unsigned long long> inc(h->C);
    for (int k = 0; k < h->C; k++) {
        double fr = cfg->center_hz[k] / cfg->fs_in;
        if (!(std::fabs(fr) < 0.5)) { delete h; return chan_fail(nullptr, FMD_ERR_ARG, "station %d centre %g Hz is outside +-fs_in/2", k, cfg->center_hz[k]); }
        fr -= std::floor(fr);                                       // [0, 1) turns per sample
        inc[k] = (unsigned long long)std::llround(std::ldexp(fr, 63)) << 1;   // fr * 2^64, even
    }
    bool ok = hipSetDevice(dev) == hipSuccess;
    for (int i = 0; i < 2; i++) ok = ok && hipMalloc(&h->win[i], sizeof(float2) * (size_t)T) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&h->done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(&h->taps, sizeof(float) * h->h_taps.size()) == hipSuccess;
    ok = ok && hipMalloc(&h->inc, sizeof(unsigned long long) * h->C) == hipSuccess;
    for (int i = 0; i < 2; i++) ok = ok && hipMemset(h->win[i], 0, sizeof(float2) * (size_t)T) == hipSuccess;
    ok = ok && hipMemcpy(h->taps, h->h_taps.data(), sizeof(float) * h->h_taps.size(), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(h->inc, inc.data(), sizeof(unsigned long long) * h->C, hipMemcpyHostToDevice) == hipSuccess;
    // the matrix-core form: L == 16 and the operand / window sizes it is built for (10 MSa/s -> 256 kSa/s with 640 taps per phase)
    if (ok && L == 16 && (15 * M) / 16 + T <= kMK && (kMG - 1) * M + kMK <= kMRail && !fmd::dev_env("FMD_CHAN_VALU")) {
        std::vector<float> at((size_t)4 * kMKW * 64, 0.0f);
        for (int w = 0; w < 4; w++)
            for (int j = 0; j < kMKW; j++)
                for (int l = 0; l < 64; l++) {
                    const int m = l & 15, tau = 4 * (kMKW * w + j) + (l >> 4);
                    const int t = (m * M) / 16 + (T - 1) - tau, p = (m * M) % 16;
                    if (t >= 0 && t < T) at[((size_t)w * kMKW + j) * 64 + l] = h->h_taps[(size_t)t * L + p];
                }
        ok = hipMalloc(&h->atab, sizeof(float) * at.size()) == hipSuccess &&
             hipMemcpy(h->atab, at.data(), sizeof(float) * at.size(), hipMemcpyHostToDevice) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(k_channelize16_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * (2 * kMRail + 4 * 256))) == hipSuccess;
    }
    if (!ok) { fmd_chan_destroy(h); return chan_fail(nullptr, FMD_ERR_DEVICE, "device allocation failed"); }
    *out = h;
    return FMD_OK;
}

int fmd_chan_destroy(fmd_channelizer h) {
    if (!h) return FMD_ERR_ARG;
    (void)hipSetDevice(h->device);
    for (int i = 0; i < 2; i++) if (h->win[i]) (void)hipFree(h->win[i]);
    if (h->done) (void)hipEventDestroy(h->done);
    if (h->taps) (void)hipFree(h->taps);
    if (h->atab) (void)hipFree(h->atab);
    if (h->inc) (void)hipFree(h->inc);
    delete h;
    return FMD_OK;
}

int fmd_chan_info(fmd_channelizer h, int* L, int* M, int* taps_per_phase, int* n_stations) {
    if (!h) return FMD_ERR_ARG;
    if (L) *L = h->L;
    if (M) *M = h->M;
    if (taps_per_phase) *taps_per_phase = h->T;
    if (n_stations) *n_stations = h->C;
    return FMD_OK;
}

int fmd_chan_get_taps(fmd_channelizer h, float* taps, size_t cap_floats) {
    if (!h || !taps || cap_floats < h->h_taps.size()) return FMD_ERR_ARG;
    std::memcpy(taps, h->h_taps.data(), sizeof(float) * h->h_taps.size());
    return FMD_OK;
}

int fmd_chan_reset(fmd_channelizer h) {
    if (!h) return FMD_ERR_ARG;
    if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return chan_fail(h, FMD_ERR_DEVICE, "synchronise failed");
    for (int i = 0; i < 2; i++)
        if (hipMemset(h->win[i], 0, sizeof(float2) * (size_t)h->T) != hipSuccess) return chan_fail(h, FMD_ERR_DEVICE, "memset failed");
    h->n_abs = 0; h->o_abs = 0; h->cur = 0; h->have_done = false;
    return FMD_OK;
}

int fmd_chan_process_cf32_dev(fmd_channelizer h, const float* d_wide, size_t n_in, float* d_out, size_t out_capacity_per_station, size_t* n_out, void* stream) {
    if (!h || !d_wide || !d_out || !n_out) return FMD_ERR_ARG;
    if (n_in == 0 || n_in > h->max_in) return chan_fail(h, FMD_ERR_SIZE, "n_in %zu outside (0, %zu]", n_in, h->max_in);
    if ((n_in * (size_t)h->L) % (size_t)h->M != 0) return chan_fail(h, FMD_ERR_SIZE, "n_in must be a multiple of %d so that a whole number of output samples results", h->M / std::gcd(h->L, h->M));
    const size_t no = n_in * (size_t)h->L / (size_t)h->M;
    if (no > out_capacity_per_station) return chan_fail(h, FMD_ERR_SIZE, "output capacity %zu < %zu", out_capacity_per_station, no);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipSetDevice(h->device) != hipSuccess) return chan_fail(h, FMD_ERR_DEVICE, "hipSetDevice failed");
    const int T = h->T;
    // the windows carry state from call to call: a caller that switches streams is ordered behind the previous call's work
    if (h->have_done && hipStreamWaitEvent(s, h->done, 0) != hipSuccess) return chan_fail(h, FMD_ERR_DEVICE, "stream wait failed");
    // window = [T-1 history samples][this block], read in place; its first sample has absolute input index n_abs - (T-1)
    const ChanWin win{h->win[h->cur], reinterpret_cast<const float2*>(d_wide), h->win[h->cur ^ 1]};
    ChanDims d{h->L, h->M, T, h->C, (long long)no, (long long)out_capacity_per_station, h->o_abs, h->n_abs - (unsigned long long)(T - 1), (long long)n_in};
    // outputs o0 .. o0+no-1 need inputs up to floor((o0+no-1) M / L) <= n_abs + n_in - 1 by construction
    if (h->atab) {
        // every workgroup keeps its operand registers over several tiles
        const int n_tiles = (int)((no + kTile - 1) / kTile);
        // two workgroups per CU: three (what its 164 registers and 49 KB of LDS allow) leave no wavefront slot and no LDS for the
        // demodulator's kernels that run beside it, which then wait for a 60 us persistent workgroup to finish — the configs[4] line x537
        // with three, x571-581 with two (profiles/round5/rds_stage_ab.txt)
        int per_cu = 2;
        if (const char* e = fmd::dev_env("FMD_CHAN_WG_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 3) per_cu = v; }      // (development A/B)
        int gx = (per_cu * 256) / h->C;
        gx = gx < 1 ? 1 : (gx > n_tiles ? n_tiles : gx);
        hipLaunchKernelGGL(k_channelize16_mfma, dim3((unsigned)gx, (unsigned)h->C), dim3(256), sizeof(float) * (2 * kMRail + 4 * 256), s, d, win, h->atab, h->inc,
                           reinterpret_cast<float2*>(d_out), n_tiles);
    } else if (h->L == 16 && T % 64 == 0 && T <= 1024)
        hipLaunchKernelGGL(k_channelize16, dim3((unsigned)((no + kTile - 1) / kTile), (unsigned)h->C), dim3(256), 0, s, d, win, h->taps, h->inc,
                           reinterpret_cast<float2*>(d_out));
    else
        hipLaunchKernelGGL(k_channelize, dim3((unsigned)((no + kTile - 1) / kTile), (unsigned)h->C), dim3(256), 0, s, d, win, h->taps, h->inc,
                           reinterpret_cast<float2*>(d_out));
    if (hipGetLastError() != hipSuccess) return chan_fail(h, FMD_ERR_DEVICE, "k_channelize launch failed");
    // (the last T-1 samples of [history ++ block] are the next call's history: the launch's first workgroup has written them into the
    //  OTHER history buffer, however short the block is — n_in = 625 < T - 1 = 639 is a legal call)
    if (hipEventRecord(h->done, s) != hipSuccess) return chan_fail(h, FMD_ERR_DEVICE, "event record failed");
    h->have_done = true;
    h->cur ^= 1;
    h->n_abs += n_in; h->o_abs += no;
    *n_out = no;
    return FMD_OK;
}

const char* fmd_chan_last_error(fmd_channelizer h) { return h ? h->err.c_str() : g_chan_error.c_str(); }

}  // extern "C"
