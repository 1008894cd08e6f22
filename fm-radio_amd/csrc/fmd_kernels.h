// Device-side data model and kernel launch interface of the batched FM demodulator.
// See DESIGN.md for the pipeline; each kernel's header comment cites the reference code it replaces.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fmdemod.h"

namespace fmd {

// Coefficients shared by every channel, passed to kernels by value (kernarg -> scalar loads).
struct FrontTaps {
    float b_fm_in[64];
    float b_fm_out[64];
    float b_hilbert_odd[32];  // the 32 non-zero Hilbert taps, b[1], b[3], ..., b[63]
    float fm_gain;
};

struct RdsTaps { float b[128]; };

struct LoopCoeffs {
    float pilot_k, pilot_a0, pilot_a1;       // y = fma(a0, y[n-2], k*x[n-2]) + a1*y[n-1]
    float pll_b0, pll_b1, pll_a0;            // 1-pole loop filter, newest-last arrays: b=[b0,b1], a=[a0,1]
    float ted_b0, ted_b1, ted_a0;
    float bpsk_b0, bpsk_b1, bpsk_a0;
};

// Per-channel serial state, structure-of-arrays ([field][C]) so a wavefront's 64 lanes (64 adjacent
// channels) load and store it coalesced.  Field list = SURVEY.md A.8 / reference member variables.
enum StateField : int {
    // pilot peak IIR (two copies: one advanced by the power pass, one by the PLL pass)
    SA_X1R, SA_X1I, SA_X2R, SA_X2I, SA_Y1R, SA_Y1I, SA_Y2R, SA_Y2I,
    SB_X1R, SB_X1I, SB_X2R, SB_X2I, SB_Y1R, SB_Y1I, SB_Y2R, SB_Y2I,
    S_PILOT_POWER,                 // sum |pilot|^2 of the current block (power pass -> PLL pass)
    S_AGC_PILOT_GAIN,
    S_PLL_X1, S_PLL_Y1, S_PLL_INT, S_PLL_ERR, S_PLL_T,
    S_LMR_PHASE_CUR, S_LMR_PHASE_PREV,
    S_AGC_RDS_GAIN,
    // BPSK synchroniser (reference bpsk_synchroniser.h:40-59)
    S_B_PLL_X1, S_B_PLL_Y1, S_B_PLL_INT, S_B_PLL_ERR, S_B_MIX_T,
    S_B_ZCD_XN, S_B_COOLDOWN, S_B_TED_ERR, S_B_TED_X1, S_B_TED_Y1, S_B_TED_INT, S_B_CLOCK,
    S_B_DUMP_R, S_B_DUMP_I,
    // de-emphasis IIR
    S_DE_X1, S_DE_Y1,
    // Manchester decoder (bit-packed ints stored as float bit patterns)
    S_M_FLAGS, S_M_BUF0, S_M_BUF1, S_M_BUF2, S_M_BUF3,
    S_NUM_FIELDS
};

struct Dims {
    int C;          // channels
    int N;          // baseband samples per block
    int m;          // stage-1 decimation (1, 4, 8)
    int n_fm_in, n_fm_out, n_rds, n_audio, n_est;
    int tail_base;  // baseband (m>1) or fm_in (m==1) samples of history kept per channel
};

struct Buffers {
    // history tails, ping-pong by block parity
    float2* base_tail[2];   // [C][tail_base]
    float2* iq_tail[2];     // [C][128]   last fm_out_iq samples of the previous block
    float*  dt_tail[2];     // [C][128]   last pll_dt samples of the previous block
    float*  fo_tail[2];     // [C][64]    last fm_out samples (Hilbert FIR history, de-emphasis path)
    // streams of the current block
    float2* fm_out_iq;      // [C][n_fm_out]
    float*  fm_out;         // [C][n_fm_out]  (de-emphasis path only)
    float*  pll_dt;         // [C][n_fm_out]
    float2* rds;            // [C][n_rds]
    float*  lmr_est;        // [C][n_est]
    float*  audio;          // [C][n_audio][2]
    float*  rds_sym;        // [C][n_rds]
    float2* rds_raw_sym;    // [C][n_rds]      (KEEP_TAPS)
    int*    rds_count;      // [C]
    float*  lpr;            // [C][n_audio]    (KEEP_TAPS)
    float*  lmr;            // [C][n_audio]    (KEEP_TAPS)
    uint8_t* rds_bytes;     // [C][32]
    int*    rds_bytes_count;// [C]
    // per-channel controls
    float*  b_lpr;          // [C][128]
    float*  b_lmr;          // [C][128]
    float*  deemph;         // [C][4]  b0,b1,a0,flag
    float*  mix;            // [C][2]  audio mode (as float), stereo mix factor
    float*  state;          // [S_NUM_FIELDS][C]
};

struct LaunchCtx {
    Dims d;
    Buffers b;
    FrontTaps front;
    RdsTaps rds_taps;
    LoopCoeffs loops;
    int parity;        // block parity selecting the tails to read (write goes to parity^1)
    int keep_taps;
    int any_deemph;
};

// optional per-kernel event recording (fmd_profile_*): launch_block records events[0..n_marks) around the kernels
struct ProfileMarks {
    static constexpr int kMax = 8;
    hipEvent_t ev[kMax + 1];
    const char* name[kMax];
    int n = 0;
};

// queue one block of the hot path on `stream`
hipError_t launch_block_cf32(const LaunchCtx& ctx, const float2* d_iq, hipStream_t stream, ProfileMarks* marks = nullptr);
hipError_t launch_block_u8(const LaunchCtx& ctx, const uchar2* d_iq, hipStream_t stream, ProfileMarks* marks = nullptr);
hipError_t launch_reset_state(const LaunchCtx& ctx, hipStream_t stream);
hipError_t selftest_atan2(const float* d_y, const float* d_x, float* d_out, size_t n, hipStream_t s);
hipError_t prepare_kernels();          // one-time function attributes (dynamic LDS sizes)
int front_tail_len(int m);             // input-history samples k_front needs per channel

}  // namespace fmd
