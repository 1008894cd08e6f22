"""GPU tests (pytest -m gpu) of FMD_FLAG_FAST_MATH, the tolerance mode: cheaper arithmetic, same signal flow.

Parity bar (BASELINE.json north star): audio, L+R, L-R and RDS symbols within 1e-4 RMS of the reference, RDS bits identical.
Checked against the CPU oracle on synthetic captures (through acquisition and in lock), against the golden fixtures dumped from
the compiled reference, and at bench scale through tiling.  The exact mode's bit-identity tests are in test_gpu_parity.py.
"""
import hashlib

import numpy as np
import pytest

import oraclelib as O
import synth
from conftest import rms
from gpu_parity import lib_coeffs_to_oracle, run_gpu
from rds_groups import decode_groups

pytestmark = pytest.mark.gpu

TOL_RMS = 1e-4  # BASELINE.json north_star


@pytest.fixture(scope="module")
def pkg():
    import fmradio_loader
    p = fmradio_loader.load()
    p.load_library()
    import torch
    assert torch.cuda.is_available()
    return p


def _caps(n_ch, n, fs, seed, u8=False, **kw):
    conv = synth.to_u8 if u8 else synth.to_cf32
    return np.stack([conv(synth.fm_capture(n, fs=fs, seed=seed, channel=c, **kw)["iq"]) for c in range(n_ch)])


def test_fast_math_primitives(pkg):
    """Accuracy of the tolerance mode's arithmetic on the device, against float64: the minimax arctangent (every quadrant,
    axes, zeros) and the hardware sine / cosine in turns over the phase ranges the chain feeds them."""
    rng = np.random.default_rng(1)
    n = 2_000_000
    y = np.concatenate([rng.uniform(-200, 200, n), rng.standard_normal(n) * 1e-3, rng.integers(-127, 129, n).astype(np.float64)]).astype(np.float32)
    x = np.concatenate([rng.uniform(-200, 200, n), rng.standard_normal(n) * 1e-3, rng.integers(-127, 129, n).astype(np.float64)]).astype(np.float32)
    got = pkg.selftest_fast_math("atan2", y, x)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    both_zero = (x == 0) & (y == 0)
    err = np.abs(got.astype(np.float64) - want)
    err = np.minimum(err, np.abs(err - 2 * np.pi))          # -pi and +pi are the same angle
    assert err[~both_zero].max() < 5e-7, err[~both_zero].max()
    assert np.all(got[both_zero] == 0.0)                     # atan2(+0, +0) = 0, as libm
    # the discriminator's / pilot phase detector's form: in turns, six coefficients (2.8e-7 turns by design)
    got_t = pkg.selftest_fast_math("atan2_turns", y, x)
    err_t = np.abs(got_t.astype(np.float64) - want / (2 * np.pi))
    err_t = np.minimum(err_t, np.abs(err_t - 1.0))
    assert err_t[~both_zero].max() < 3.5e-7, err_t[~both_zero].max()
    assert np.all(got_t[both_zero] == 0.0)
    t = np.concatenate([rng.uniform(-0.5, 0.5, n), rng.uniform(-2.0, 2.0, n), np.array([0.0, 0.25, -0.25, 0.5, -0.5, 1e-6, -1e-6])]).astype(np.float32)
    for kind, f in (("sin_turns", np.sin), ("cos_turns", np.cos)):
        got = pkg.selftest_fast_math(kind, t)
        want = f(2 * np.pi * t.astype(np.float64))
        e = np.abs(got.astype(np.float64) - want).max()
        assert e < 2e-6, (kind, e)


def rds_bits(byte_stream: np.ndarray) -> np.ndarray:
    """The Manchester decoder's bit stream (MSB first, as the reference packs it)."""
    return np.unpackbits(np.asarray(byte_stream, np.uint8))


def same_bits_once_in_lock(a: np.ndarray, b: np.ndarray, skip_bits: int, max_shift: int = 24) -> bool:
    """RDS parity of the tolerance mode ("RDS needs the post-lock definition"): while the BPSK synchroniser
    acquires, its zero-crossing / clock-wrap decisions sit on margins that a last-bits difference in its input can tip (two
    builds of the reference itself part ways there), so the streams may differ by a few symbols early on; from
    lock on they must carry the SAME bits.  True if, after `skip_bits`, one stream equals the other shifted by <= max_shift."""
    a, b = rds_bits(a), rds_bits(b)
    n = min(a.size, b.size) - skip_bits - max_shift
    if n < 256:
        return False
    for sh in range(-max_shift, max_shift + 1):
        if np.array_equal(a[skip_bits:skip_bits + n], b[skip_bits + sh:skip_bits + sh + n]):
            return True
    return False


def soft_symbol_stats(sym_g, cnt_g, sym_o, cnt_o, lock_blocks):
    """The tolerance mode's soft RDS symbols (OnRDSOut's payload, reference broadcast_fm_demod.cpp:327) against the oracle's, for one station.
    The two symbol streams are aligned from the first block behind which both runs emit the same number of symbols in EVERY block
    (at least `lock_blocks` in): before that a zero-crossing / clock-wrap decision of the synchroniser (bpsk_synchroniser.cpp:159-183) that
    fell on the other side has the two clocks a sample apart.  Such decisions also tip in lock now and then — two builds of the REFERENCE
    differ the same way, far more often (profiles/round4/reference_flip_evidence.json: 1.4 % of its symbols move by ~0.1-0.3 between its own
    gcc preset and a scalar build; every bit stays) — so the statistics separate the symbols that moved by more than 1e-2 from the rest.
    Returns None when fewer than 200 symbols can be compared."""
    cnt_g = np.asarray(cnt_g).reshape(-1); cnt_o = np.asarray(cnt_o).reshape(-1)[:cnt_g.size]
    agree_from = 0
    for b in range(cnt_g.size - 1, -1, -1):
        if cnt_g[b] != cnt_o[b]:
            agree_from = b + 1
            break
    first = max(agree_from, lock_blocks)
    if first >= cnt_g.size:
        return None
    lg, lo = int(cnt_g[:first].sum()), int(cnt_o[:first].sum())
    m = int(cnt_g[first:].sum())
    if m < 200:
        return None
    d = np.abs(np.asarray(sym_g, np.float64).reshape(-1)[lg:lg + m] - np.asarray(sym_o, np.float64).reshape(-1)[lo:lo + m])
    big = d > 1e-2
    return {"compared_from_block": int(first), "symbols": m, "median": float(np.median(d)), "p99": float(np.percentile(d, 99)),
            "rms": float(np.sqrt(np.mean(d ** 2))), "moved_over_1e-2": int(big.sum()),
            "rms_of_the_rest": float(np.sqrt(np.mean(d[~big] ** 2))) if (~big).any() else 0.0,
            "p99_of_the_rest": float(np.percentile(d[~big], 99)) if (~big).any() else 0.0}


def record_parity_metrics(name, metrics):
    """Merge one test's measured parity figures into gpurun_out/parity_metrics.json (FMD_PARITY_METRICS overrides the path); the builder
    copies it to profiles/roundN/parity_metrics.json.  Never fails a test."""
    import json, os, pathlib
    try:
        root = pathlib.Path(__file__).resolve().parent.parent
        path = pathlib.Path(os.environ.get("FMD_PARITY_METRICS", root / "gpurun_out" / "parity_metrics.json"))
        path.parent.mkdir(parents=True, exist_ok=True)
        doc = json.loads(path.read_text()) if path.exists() else {}
        doc[name] = metrics
        path.write_text(json.dumps(doc, indent=1, sort_keys=True))
    except Exception as e:      # noqa: BLE001
        print("parity metrics not recorded:", e)


def lmr_audio_excess(g, o, c, nb):
    """The L-R and audio bar of the tolerance mode, block by block.  Every block must be within TOL_RMS — audio too: it carries
    2 (L+R +- L-R), measured 2.4e-5 — of the oracle, except for what ONE documented discontinuity of the reference explains: each of a block's
    L-R phase estimates is +-pi/2 - atan2(im, re) by the SIGN of an L-R sample (reference broadcast_fm_demod.cpp:500-510), so a
    sample within the arithmetic difference of zero (~3e-6) lands pi away in one of two evaluations and moves the block's offset by
    0.1 pi / n_est; the next blocks' L-R are rotated by the difference of the two offsets (consumed as turns) until the tracker
    has pulled them together.  The offsets are outputs (GetAudioLMRPhaseError), so the allowance is computed from their measured
    difference: |error| <= |L-R quadrature| x 2 pi x |offset difference| ~ 0.7 x |offset difference| (audio: twice that).  (Rounds 3-5 allowed
    audio 2 x TOL_RMS without any flip: VERDICT r5 weak 2.)
    Returns (worst error / allowance over the blocks, number of flipped estimates, whole-run RMS of L-R, of audio)."""
    off_g = np.asarray(g["lmr_phase"][c], np.float64).reshape(-1)[:nb]
    off_o = o["lmr_phase"].reshape(-1)[:nb].astype(np.float64)
    doff = np.abs(off_g - off_o)
    prev = np.concatenate([[0.0], doff[:-1]])                  # the offset a block is mixed with is the one the previous block left
    flips = int(np.sum(np.abs(np.diff(np.concatenate([[0.0], off_g - off_o]))) > 7e-4))
    worst, whole = 0.0, {}
    for k, scale in (("lmr", 1.0), ("audio", 2.0)):
        d = np.asarray(g[k][c], np.float64).reshape(nb, -1) - o[k].reshape(nb, -1)
        per_block = np.sqrt((d ** 2).mean(axis=1))
        allow = np.maximum(TOL_RMS, scale * 0.7 * prev)
        worst = max(worst, float(np.max(per_block / allow)))
        whole[k] = float(np.sqrt((d ** 2).mean()))
    return worst, flips, whole["lmr"], whole["audio"]


def _compare(pkg, caps, bs, fs, from_block=0, sym_skip_s=0.4, **kw):
    """Fast mode on the GPU vs the oracle (handed the library's coefficients), per channel: worst RMS error per stream over the
    blocks from `from_block` on, and whether counts / bytes are identical."""
    g = run_gpu(pkg, caps, bs, fs, fast_math=True, **kw)
    u8 = caps.dtype == np.uint8
    m = fs // 256_000
    n_fm_out = bs // m // 2
    worst = {k: 0.0 for k in ("audio", "lpr", "lmr", "fm_out_iq", "pll_dt", "rds_sym", "rds_sym_rest_rms", "lmr_audio_excess", "flips", "sym_stations", "sym_moved", "sym_total", "same_counts_other_bytes")}
    counts_equal = bytes_equal = 0
    bits_equal = True
    for c in range(caps.shape[0]):
        o = O.run_chain(caps[c], bs, fs, u8=u8, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]),
                        streams=["fm_out_iq", "pll_dt", "lpr", "lmr", "audio", "rds_sym", "lmr_phase"])
        ex, fl, _, _ = lmr_audio_excess(g, o, c, caps.shape[1] // bs)
        worst["lmr_audio_excess"] = max(worst["lmr_audio_excess"], ex); worst["flips"] += fl
        per = {"audio": 2 * n_fm_out // 4, "lpr": n_fm_out // 4, "lmr": n_fm_out // 4, "fm_out_iq": 2 * n_fm_out, "pll_dt": n_fm_out}
        for k, w in per.items():
            a = np.asarray(g[k][c], np.float64).reshape(-1)[from_block * w:]
            b = o[k].reshape(-1).astype(np.float64)[from_block * w:]
            if k == "pll_dt":                                 # phases in turns: compare modulo 1
                dlt = a - b
                dlt -= np.round(dlt)
                worst[k] = max(worst[k], rms(dlt))
            else:
                worst[k] = max(worst[k], rms(a - b))
        if np.array_equal(g["rds_count"][c], o["rds_count"]):
            counts_equal += 1
            # ... then the decoder saw the same number of symbols at every block edge: no shift is possible, and whatever differs is a soft symbol
            # within the arithmetic difference of zero while the loops pull in — it must lie in front of the lock point (5 groups of 104 bits)
            gb, ob = np.unpackbits(g["rds_bytes"][c]), np.unpackbits(o["rds_bytes"])
            nz = np.nonzero(gb != ob)[0] if gb.size == ob.size else np.array([1 << 30])
            worst["same_counts_other_bytes"] += int(nz.size > 0 and (nz[-1] >= 5 * 104 or nz.size > 8))
        # symbol VALUES from lock on, for the stations whose symbol clock agrees with the oracle's from the first block (a run of 12 blocks is
        # too short to wait for two clocks that settled a sample apart to be pulled together: tests/test_gpu_long.py compares every station)
        st = soft_symbol_stats(g["rds_sym"][c], g["rds_count"][c], o["rds_sym"], o["rds_count"], max(from_block, int(np.ceil(sym_skip_s * fs / bs))))
        if st is not None and np.array_equal(g["rds_count"][c], o["rds_count"]):
            worst["sym_stations"] += 1
            worst["rds_sym"] = max(worst["rds_sym"], st["median"])
            worst["rds_sym_rest_rms"] = max(worst["rds_sym_rest_rms"], st["rms_of_the_rest"])
            worst["sym_moved"] += st["moved_over_1e-2"]; worst["sym_total"] += st["symbols"]
        bytes_equal += int(np.array_equal(g["rds_bytes"][c], o["rds_bytes"]))
        bits_equal = bits_equal and same_bits_once_in_lock(g["rds_bytes"][c], o["rds_bytes"], skip_bits=5 * 76)
    return worst, counts_equal, bytes_equal, bits_equal


@pytest.mark.parametrize("fs,u8", [(256_000, False), (256_000, True), (1_024_000, False), (1_024_000, True), (2_048_000, False), (2_048_000, True)])
def test_fast_mode_is_within_the_north_star_tolerance_of_the_oracle(pkg, fs, u8):
    """Every block from the very first (acquisition included): audio, L+R, L-R and the discriminator output within 1e-4 RMS,
    RDS bits identical once the synchroniser is in lock (same_bits_once_in_lock), and for most stations from the first
    byte.  RDS symbol VALUES: 99 % of them within 2e-3 of the symbol RMS: with a synthetic pilot
    at exactly 19 000 Hz the reference's NCO frequency word rounds to exactly -19000 for every control value within +-1e-5, a
    dead zone inside which its phase error drifts freely (a relaxation cycle of ~1e-4 turns); two evaluations that differ in
    the last bits leave the dead zone at different samples and differ by ~1e-5 turns of pilot phase for a while, which the
    x3 harmonic turns into ~2e-4 of the 57 kHz subcarrier."""
    bs = fs * 64 // 1000
    caps = _caps(5, 12 * bs, float(fs), seed=9100 + (1 if u8 else 0), u8=u8)
    worst, counts_equal, bytes_equal, bits_equal = _compare(pkg, caps, bs, fs)
    print("fast-vs-oracle worst RMS:", {k: f"{v:.2e}" for k, v in worst.items()}, "stations with identical symbol counts / bytes:", counts_equal, bytes_equal, "of 5")
    assert bits_equal
    assert bytes_equal >= 3
    # a station whose synchroniser emitted the oracle's symbol COUNTS block for block emitted its bits too, bit for bit and unshifted, from the lock
    # point on, and differs in at most 8 acquisition bits in front of it — every such station (the others' clocks settled a sample apart during
    # acquisition; their bits are compared from lock on, up to a shift, above)
    assert worst["same_counts_other_bytes"] == 0, worst
    # numeric ceilings within 2x of what profiles/round5/parity_metrics.json records (the north star's own figure is 1e-4)
    for k in ("lpr", "fm_out_iq"):
        assert worst[k] <= 2e-6, (k, worst[k])
    assert worst["lmr_audio_excess"] <= 1.0, worst     # L-R and audio: every block within 1e-4 (lmr_audio_excess: the one allowance and why)
    # soft symbols on at least 3 of the 5 stations, over the 0.35 s this short run has behind the synchroniser's lock (within 2e-3 of the symbol
    # RMS, 0.7); the bounds of a settled synchroniser (median 1e-4, typical station 2e-4 RMS, no more moved symbols than the reference's own
    # builds) are asserted on 24 stations x 2 s, every station compared, in tests/test_gpu_long.py
    assert worst["sym_stations"] >= 3, worst
    assert worst["rds_sym"] <= 1.5e-4 and worst["rds_sym_rest_rms"] <= 5.5e-4, worst      # (measured: 7.4e-5 / 2.7e-4 on the worst configuration)
    assert worst["sym_moved"] <= 0.003 * worst["sym_total"], worst                          # (measured: 5 of 3805)
    assert worst["pll_dt"] <= 5e-5          # turns
    record_parity_metrics(f"fast_vs_oracle_12_blocks_fs{fs}_{'u8' if u8 else 'cf32'}", {k: float(v) for k, v in worst.items()})


@pytest.mark.parametrize("fs,u8", [(1_024_000, False), (1_024_000, True), (2_048_000, False), (2_048_000, True)])
def test_fast_mode_fused_first_decimator_is_bit_identical_to_the_two_kernels(pkg, fs, u8):
    """k_front_pre_mfma (first decimator + front end in one kernel, fm_in in LDS) against k_predecim_mfma + k_front_mfma with fm_in
    through HBM (fmd_debug_split_front): the same arithmetic per phase and per fm_out sample, so EVERY output stream is bit-identical,
    over several blocks (the histories both forms keep: 64 input samples, 319 phases) and for one- and many-tile blocks."""
    for bs in (fs * 64 // 1000, 8192 * (fs // 256_000)):        # 8 tiles of 1024 fm_out samples / one tile
        nb = 5
        caps = _caps(3, nb * bs, float(fs), seed=9400 + (1 if u8 else 0), u8=u8)
        a = run_gpu(pkg, caps, bs, fs, fast_math=True)
        b = run_gpu(pkg, caps, bs, fs, fast_math=True, split_front=True)
        for k in a:
            if k == "coeffs":
                continue
            if isinstance(a[k], np.ndarray):
                assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (k, bs)
            else:
                for x, y in zip(a[k], b[k]):
                    assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8)), (k, bs)


def test_fast_mode_first_decimator_small_tile_and_format_switch(pkg):
    """k_predecim_mfma's 256-output tiles (u8 blocks that are no multiple of its 2048-output tile: 12288 samples at 1.024 MSa/s), and a
    handle fed u8 blocks and cf32 blocks in turn (the first decimator's history is kept as cf32 whatever the capture's format):
    discriminator output and L+R within 1e-4 RMS of the oracle."""
    fs, bs, nb = 1_024_000, 12288, 48
    caps = _caps(3, nb * bs, float(fs), seed=9300, u8=True)
    worst, _, _, _ = _compare(pkg, caps, bs, fs)
    print("small tiles:", {k: f"{v:.2e}" for k, v in worst.items()})
    for k in ("lpr", "fm_out_iq"):
        assert worst[k] <= TOL_RMS, (k, worst[k])
    # u8 and cf32 blocks in turn: the cf32 blocks carry the same integers, so the oracle's u8 run is the reference for both
    bs = 65536
    caps = _caps(2, 6 * bs, float(fs), seed=9301, u8=True)
    dm = pkg.BatchDemod(2, bs, fs, keep_taps=True, fast_math=True)
    coeffs = [dm.get_coeffs(c) for c in range(2)]
    lpr = []
    for b in range(6):
        blk = np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])
        if b % 2:
            blk = blk.astype(np.float32) - 127.0                      # (reference src/app.cpp:56-62)
        assert dm.process(blk) == 0
        lpr.append(dm.stream("lpr").copy())
    dm.close()
    lpr = np.concatenate(lpr, axis=1)
    for c in range(2):
        o = O.run_chain(caps[c], bs, fs, u8=True, coeffs=lib_coeffs_to_oracle(coeffs[c]), streams=["lpr"])
        assert rms(lpr[c].astype(np.float64) - o["lpr"].reshape(-1)) <= TOL_RMS


def test_fast_mode_blocks_longer_than_the_inline_lmr_phase_limit(pkg):
    """Up to 5120 audio samples per block the next block's k_extract integrates the L-R phase estimates itself; longer blocks
    (here 65536 samples at 256 kSa/s = 8192 audio samples, 820 estimates) take the k_lmr_phase kernel behind k_extract.

    Also shows the one discontinuity on the audio path (reference :500-510): each estimate is +-pi/2 - atan2(im, re) by the
    SIGN of the L-R sample, so a sample within the arithmetic difference of zero (here ~1e-5) lands pi away in one of the two
    evaluations and moves that block's offset by 0.1 pi / n_est.  The L-R of the next block is then rotated by that much (still
    < 1e-3 RMS) until the loop has pulled the two back together; L+R is never affected."""
    bs, nb, fs = 65536, 8, 256_000
    caps = _caps(2, nb * bs, float(fs), seed=9300)
    g = run_gpu(pkg, caps, bs, fs, fast_math=True)
    flips = 0
    for c in range(2):
        o = O.run_chain(caps[c], bs, fs, u8=False, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["lpr", "lmr", "audio", "lmr_phase", "rds_sym"])
        assert rms(np.asarray(g["lpr"][c], np.float64).reshape(-1) - o["lpr"].reshape(-1)) <= TOL_RMS
        ex, fl, _, _ = lmr_audio_excess(g, o, c, nb)
        flips += fl
        assert ex <= 1.0, (c, ex)
        assert same_bits_once_in_lock(g["rds_bytes"][c], o["rds_bytes"], skip_bits=5 * 76)
    print("flipped L-R phase estimates:", flips)


@pytest.mark.parametrize("fs,bs", [(256_000, 16384), (1_024_000, 32768), (256_000, 10240)])
def test_fast_mode_de_emphasis_inside_the_front_tile(pkg, fs, bs):
    """Tolerance mode runs the de-emphasis IIR (reference :403-406) inside k_front's tile from a zero state 128 samples early
    (time constants up to ~79 us) instead of as a serial stage of its own: 50 us, 75 us and unfiltered channels in one batch,
    and a 150 us channel that makes the whole handle fall back to the serial k_deemphasis + k_hilbert stage."""
    from fm_radio_amd.capi import default_controls
    from gpu_parity import oracle_controls

    def ctl(**kw):
        c = default_controls()
        for k, v in kw.items():
            setattr(c, k, v)
        return c
    nb = -(-fs * 12 // 10 // bs)           # 1.2 s: enough RDS bits behind the synchroniser's acquisition (slower on the attenuated subcarrier)
    caps = _caps(4, nb * bs, float(fs), seed=9500)
    for per in ({0: ctl(use_deemphasis=1, deemphasis_tus=50), 1: ctl(use_deemphasis=1, deemphasis_tus=75), 3: ctl(use_deemphasis=1, deemphasis_tus=50, audio_out=1)},
                {0: ctl(use_deemphasis=1, deemphasis_tus=50), 2: ctl(use_deemphasis=1, deemphasis_tus=150)}):
        g = run_gpu(pkg, caps, bs, fs, fast_math=True, per_channel_controls=per)
        for c in range(4):
            o = O.run_chain(caps[c], bs, fs, u8=False, controls=oracle_controls(per[c]) if c in per else None,
                            coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["fm_out_iq", "lpr", "lmr", "audio", "rds_sym"])
            for k in ("fm_out_iq", "lpr", "lmr", "audio"):
                e = rms(np.asarray(g[k][c], np.float64).reshape(-1) - o[k].reshape(-1))
                assert e <= TOL_RMS, (sorted(per), c, k, e)
            assert same_bits_once_in_lock(g["rds_bytes"][c], o["rds_bytes"], skip_bits=12 * 76), (sorted(per), c)   # (a symbol slipped during acquisition re-pairs the Manchester decoder 0.5 s later)


@pytest.mark.parametrize("fs,bs", [(256_000, 16384), (1_024_000, 32768)])
def test_fast_mode_de_emphasis_switched_on_and_off_between_blocks(pkg, fs, bs):
    """(1.024 MSa/s: the blocks with the filter on run the two kernels k_predecim_mfma + k_front_mfma, the others k_front_pre_mfma —
    both keep both histories, 64 input samples and 319 phases, so a handle changes form between any two blocks.)
    Controls take effect at the next block boundary; the in-tile filter has no state of its own, so on L+R (no loop in its
    path: the pilot PLL and the L-R phase offset, which see the filtered multiplex, take many blocks to move over) a channel that
    switches it on (or off) mid-stream is, from that block on, what a run with the filter always on (off) gives."""
    from fm_radio_amd.capi import default_controls
    nb = 6
    caps = _caps(2, nb * bs, float(fs), seed=9600)
    on = default_controls(); on.use_deemphasis = 1; on.deemphasis_tus = 50
    off = default_controls()
    mono = lambda a: a.reshape(a.shape[0], -1, 2).sum(axis=2)
    always_on = mono(run_gpu(pkg, caps, bs, fs, fast_math=True, controls=on)["audio"]).reshape(2, nb, -1)
    always_off = mono(run_gpu(pkg, caps, bs, fs, fast_math=True)["audio"]).reshape(2, nb, -1)
    dm = pkg.BatchDemod(2, bs, fs, fast_math=True)
    got = []
    for b in range(nb):
        if b == 2:
            dm.set_controls(on)
        if b == 4:
            dm.set_controls(off)
        assert dm.process(np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])) == 0
        got.append(mono(dm.audio().reshape(2, -1)))
    dm.close()
    for b in range(nb):
        want = always_on if 2 <= b < 4 else always_off
        # the block after a switch: the audio FIRs (128 taps at 128 kHz) still hold 1 ms of the other signal
        tail = slice(32, None) if b in (2, 4) else slice(None)
        assert rms((got[b] - want[:, b])[:, tail]) <= 2 * TOL_RMS, b    # (the sum of two channels)


def test_fast_mode_odd_block_length_and_per_station_cut_offs(pkg):
    """A block length whose tiles are the small ones (9216 samples at 256 kSa/s: 512-sample front tiles, 128-sample extract tiles
    on the VALU), and eleven stations with eleven different L+R / L-R cut-offs: k_extract_bp takes its tap tables (and the matrix of
    the block-edge sums) per distinct cut-off (the table grows past its first allocation here) through a per-station index."""
    from fm_radio_amd.capi import default_controls
    from gpu_parity import oracle_controls
    for bs, n_ch in ((9216, 3), (16384, 11)):
        caps = _caps(n_ch, -(-230_400 // bs) * bs, 256_000.0, seed=9700)    # 0.9 s
        per = {}
        for c in range(n_ch):
            k = default_controls()
            k.lpr_cutoff_hz = 15000 - 700 * c
            k.lmr_cutoff_hz = 14000 - 600 * c
            per[c] = k
        g = run_gpu(pkg, caps, bs, 256_000, fast_math=True, per_channel_controls=per)
        for c in range(n_ch):
            o = O.run_chain(caps[c], bs, 256_000, u8=False, controls=oracle_controls(per[c]), coeffs=lib_coeffs_to_oracle(g["coeffs"][c]),
                            streams=["fm_out_iq", "lpr", "lmr", "audio", "rds_sym", "lmr_phase"])
            for k in ("fm_out_iq", "lpr"):
                e = rms(np.asarray(g[k][c], np.float64).reshape(-1) - o[k].reshape(-1))
                assert e <= TOL_RMS, (bs, c, k, e)
            ex, _, _, _ = lmr_audio_excess(g, o, c, caps.shape[1] // bs)
            assert ex <= 1.0, (bs, c, ex)
            assert same_bits_once_in_lock(g["rds_bytes"][c], o["rds_bytes"], skip_bits=5 * 76), (bs, c)


@pytest.mark.parametrize("bs", [2048, 4096, 6144, 12288])
def test_fast_mode_short_blocks_whose_tiles_do_not_fill_a_workgroup(pkg, bs):
    """k_extract_bp gives each of a workgroup's four wavefronts tiles of 256 audio samples (tile_first + w, + 4, ...): blocks of 1, 2, 3 and
    6 tiles leave wavefronts without a tile or with fewer than the others; the block edge (the first 31 outputs' sums over the previous
    block) then comes round every 1-6 tiles.  L+R, L-R and audio against the oracle from the first block."""
    nb = 8 * 16384 // bs
    caps = _caps(2, nb * bs, 256_000.0, seed=9500 + bs)
    g = run_gpu(pkg, caps, bs, 256_000, fast_math=True)
    for c in range(2):
        o = O.run_chain(caps[c], bs, 256_000, u8=False, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["lpr", "lmr", "audio", "lmr_phase"])
        assert rms(g["lpr"][c].astype(np.float64) - o["lpr"]) <= TOL_RMS, (bs, c)
        assert lmr_audio_excess(g, o, c, nb)[0] <= 1.0, (bs, c)
        assert np.all(np.isfinite(g["audio"][c]))


def test_fast_mode_golden_chain_fixture(pkg, golden):
    """Against vectors dumped from the compiled reference (tests/golden/chain_b16384.npz), the same bar the exact mode meets."""
    g = golden("chain_b16384.npz")
    out = run_gpu(pkg, g["capture"][None], 16384, 1_024_000, fast_math=True)
    assert rms(out["audio"][0].reshape(-1) - g["audio"]) <= TOL_RMS
    assert rms(out["lmr"][0] - g["lmr"]) <= TOL_RMS
    assert rms(out["lpr"][0] - g["lpr"]) <= TOL_RMS
    assert rms(out["fm_out_iq"][0] - g["fm_out_iq"]) <= TOL_RMS
    assert np.array_equal(out["rds_count"][0], g["rds_count"])
    assert rms(out["rds_sym"][0] - g["rds_sym"]) <= 1e-3 * 0.7
    assert np.array_equal(out["rds_bytes"][0], g["rds_bytes"])


def test_fast_mode_long_run_rds_known_answer(pkg, golden):
    """2.6 s at block 65536 against the reference's dumped RDS stream: the same bits from lock on, the synthesised groups decode."""
    g = golden("long_b65536.npz")
    nb, bs, seed = int(g["n_blocks"]), int(g["block_size"]), int(g["seed"])
    cap = synth.to_u8(synth.fm_capture(nb * bs, seed=seed)["iq"])
    if hashlib.sha256(cap.tobytes()).hexdigest() != str(g["capture_sha256"]):
        pytest.skip("synthetic capture not bit-reproducible with this numpy build")
    out = run_gpu(pkg, cap[None], bs, 1_024_000, fast_math=True)
    assert same_bits_once_in_lock(out["rds_bytes"][0], g["rds_bytes"], skip_bits=2 * 608)   # RDS bits: identical from lock on
    assert abs(int(out["rds_count"][0].sum()) - int(g["rds_count"].sum())) <= 8
    audio = out["audio"][0].reshape(nb, -1)
    for i, b in enumerate(g["audio_blocks"]):
        assert rms(audio[int(b)] - g["audio"][i]) <= TOL_RMS
    got = decode_groups(out["rds_bytes"][0])
    want = {tuple(int(v) for v in w) for w in g["groups"]}
    assert len(got) >= 20 and sum(1 for w in got if w in want) >= len(got) - 1


def test_fast_mode_with_detuned_noisy_and_missing_pilots(pkg):
    """Stations the pilot PLL cannot hold (pilot 130 Hz off, 30 Hz outside the loop's range: saturated control and integrator;
    weak pilot under heavy noise; no pilot at all), next to a normal one.  The weak pilot is tracked within the audio tolerance.
    A loop 30 Hz out of range never locks: its phase error sweeps through 2 pi thirty times a second and the trajectory is that of
    a driven nonlinear oscillator, on which differences grow; the control enters and leaves its rail twice per beat and the span in which
    it does is evaluated with the linear model (fmd_kernels_fast.inc k_pll_span): ~1e-3 turns of NCO phase, 2e-3 RMS on audio, after
    0.6 s; the stereo image of such a station is meaningless in the reference too - bounded at 3e-3 here.  The
    pilot-less station's L-R is demodulated noise, compared on L+R only."""
    n = 10 * 16384
    caps = np.stack([
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=501, channel=0)["iq"]),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=502, channel=1, pilot_hz=19130.0)["iq"]),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=504, channel=3, pilot_level=0.02, noise_sigma=0.3)["iq"]),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=505, channel=4, pilot_hz=18870.0)["iq"]),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=503, channel=2, pilot_level=0.0)["iq"]),
    ])
    worst, _, _, _ = _compare(pkg, caps[:1], 16384, 256_000)
    assert worst["lmr_audio_excess"] <= 1.0, worst
    g = run_gpu(pkg, caps, 16384, 256_000, fast_math=True)
    for c in range(5):
        o = O.run_chain(caps[c], 16384, 256_000, u8=False, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["lpr", "lmr", "audio", "pll_dt", "lmr_phase"])
        assert rms(g["lpr"][c].astype(np.float64) - o["lpr"]) <= TOL_RMS, c
        e_audio = rms(g["audio"][c].reshape(-1).astype(np.float64) - o["audio"].reshape(-1))
        dlt = g["pll_dt"][c].astype(np.float64) - o["pll_dt"]
        dlt -= np.round(dlt)
        print(f"channel {c}: audio rms err {e_audio:.2e}, pll phase rms err {rms(dlt):.2e} turns")
        if c in (1, 3):   # (an unlocked loop has no restoring force: what a span's approximation leaves in the NCO phase stays there)
            assert e_audio <= 3e-3, (c, e_audio)
        elif c < 4:
            assert lmr_audio_excess(g, o, c, 10)[0] <= 1.0, (c, e_audio)


def test_fast_mode_is_deterministic_and_batch_independent(pkg):
    """A station's outputs do not depend on the batch it is in or on its position in it, and repeat exactly."""
    import torch
    bs, nb = 16384, 6
    base = _caps(4, nb * bs, 256_000.0, seed=9300)
    small = run_gpu(pkg, base, bs, 256_000, fast_math=True)
    again = run_gpu(pkg, base, bs, 256_000, fast_math=True)
    assert np.array_equal(small["audio"].view(np.uint32), again["audio"].view(np.uint32))
    n_ch = 1024 + 3
    idx = np.arange(n_ch) % 4
    dbase = torch.from_numpy(base).cuda()
    tidx = torch.from_numpy(idx).cuda()
    dm = pkg.BatchDemod(n_ch, bs, 256_000, fast_math=True)
    for b in range(nb):
        dm.process(dbase[:, b * bs:(b + 1) * bs][tidx].contiguous())
    audio = dm.audio()
    by, bc = dm.rds_bytes()
    want = small["audio"][:, -audio.shape[1] * 2:].reshape(4, -1, 2)
    assert np.array_equal(audio.view(np.uint32), want[idx].view(np.uint32))
    assert np.all(np.isfinite(audio))
    dm.close()


def test_fast_mode_dead_channel_cannot_slow_or_disturb_its_neighbours(pkg):
    """All-zero input (pilot AGC 1/0 -> NaN, as in the reference) and noise-only channels beside healthy ones: the healthy
    channels' outputs are what they are alone, and the PLL kernel's spans stay full (cost independent of lock)."""
    bs, nb = 16384, 6
    good = _caps(2, nb * bs, 256_000.0, seed=9400)
    rng = np.random.default_rng(5)
    noise = (2.0 * rng.standard_normal((nb * bs, 2))).astype(np.float32)
    zero = np.zeros((nb * bs, 2), np.float32)
    caps = np.stack([good[0], zero, noise, good[1]])
    alone = run_gpu(pkg, good, bs, 256_000, fast_math=True)
    mixed = run_gpu(pkg, caps, bs, 256_000, fast_math=True)
    for i, c in enumerate((0, 3)):
        assert np.array_equal(alone["audio"][i].view(np.uint32), mixed["audio"][c].view(np.uint32))
        assert np.array_equal(alone["rds_bytes"][i], mixed["rds_bytes"][c])
    assert np.all(mixed["audio"][1].reshape(nb, -1)[-1] == 0.0)               # a dead input demodulates to silence, as in the reference
    dm = pkg.BatchDemod(4, bs, 256_000, fast_math=True)
    for b in range(nb):
        dm.process(caps[:, b * bs:(b + 1) * bs])
    st = dm.spec_stats()["pll"]
    dm.close()
    assert st["samples_per_span"] > 15.0, st


def test_one_launch_form_of_a_steady_block_matches_the_three_launch_form(pkg):
    """k_chain (fmd_debug_set_chain, fm-radio_amd/csrc/fmd_kernels_chain.inc: front end, pilot stage and extract stage of a steady 256 kSa/s cf32
    block as ONE launch, fm_out in LDS) against the three-launch form that is the default and against the oracle: the same distance from the
    oracle, a few 1e-6 from each other (its discriminator takes one arctangent of x[n] conj(x[n-1]) where the other takes two arctangents'
    difference), identical RDS bits from lock on — and a handle that CHANGES between the two forms every block (they keep the same
    histories: IQ tail, fm_out tail, pilot columns, last cubic, loop state) stays as close.  11 stations: a workgroup's 8 and a ragged one."""
    import torch
    fs, bs, nb, n_ch = 256_000, 16384, 12, 11
    caps = _caps(n_ch, nb * bs, float(fs), seed=9700)

    def run(schedule):
        dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
        au, by = [], [b"" for _ in range(n_ch)]
        for b in range(nb):
            dm.set_chain(schedule(b))
            assert dm.submit(torch.from_numpy(np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])).cuda()) == 0
            dm.synchronize()
            au.append(dm.audio().reshape(n_ch, -1).copy())
            b_, bc = dm.rds_bytes()
            for c in range(n_ch):
                by[c] += b_[c, :bc[c]].tobytes()
        k = [dm.get_coeffs(c) for c in range(n_ch)]
        n_chain = dm.chain_blocks()
        dm.close()
        return np.concatenate(au, axis=1), [np.frombuffer(x, np.uint8) for x in by], k, n_chain

    three, by3, k, n0 = run(lambda b: False)
    one, by1, _, n1 = run(lambda b: True)
    mixed, bym, _, nm = run(lambda b: b % 2 == 1)
    assert n0 == 0 and n1 == nb - 1 and nm == nb // 2      # (the first block is a start-up block: always the three launches)
    na = one.shape[1] // nb
    worst = {"one_vs_three": 0.0, "mixed_vs_three": 0.0, "one_vs_oracle": 0.0, "three_vs_oracle": 0.0}
    for c in range(n_ch):
        worst["one_vs_three"] = max(worst["one_vs_three"], rms(one[c].astype(np.float64) - three[c]))
        worst["mixed_vs_three"] = max(worst["mixed_vs_three"], rms(mixed[c].astype(np.float64) - three[c]))
        assert same_bits_once_in_lock(by1[c], by3[c], skip_bits=5 * 76) and same_bits_once_in_lock(bym[c], by3[c], skip_bits=5 * 76), c
    for c in (0, 3, 7, 8, 10):
        o = O.run_chain(caps[c], bs, fs, u8=False, coeffs=lib_coeffs_to_oracle(k[c]), streams=["audio"])["audio"].reshape(-1)
        for name, a in (("one_vs_oracle", one), ("three_vs_oracle", three)):
            per_block = [rms(a[c, b * na:(b + 1) * na].astype(np.float64) - o[b * na:(b + 1) * na]) for b in range(4, nb)]      # (in lock: no flipped estimates in play)
            worst[name] = max(worst[name], max(per_block))
        assert same_bits_once_in_lock(by1[c], O.run_chain(caps[c], bs, fs, u8=False, coeffs=lib_coeffs_to_oracle(k[c]), streams=["rds_sym"])["rds_bytes"], skip_bits=5 * 76), c
    print("k_chain:", {a: f"{b:.2e}" for a, b in worst.items()})
    assert worst["one_vs_three"] <= 5e-6 and worst["mixed_vs_three"] <= 5e-6, worst        # (measured 1.2e-6)
    assert worst["one_vs_oracle"] <= TOL_RMS and worst["one_vs_oracle"] <= 1.25 * worst["three_vs_oracle"] + 2e-6, worst
    record_parity_metrics("one_launch_form_k_chain_11_stations_12_blocks", worst)
    # Repetition on fresh handles, 64 stations (8 workgroups: beside the RDS stage's workgroup of the block before, whose LDS they inherit): the
    # kernel must not depend on what it finds in LDS — round 6 found quad 303 of a window (behind its last sample, under zero taps) unwritten:
    # 0 x a NaN pattern left by another kernel made column 15 of every tile NaN on 3 of 4 runs on some boxes and never on others
    caps64 = caps[np.arange(64) % n_ch][:, :4 * bs]
    blocks = [torch.from_numpy(np.ascontiguousarray(caps64[:, b * bs:(b + 1) * bs])).cuda() for b in range(4)]
    def short(chain):
        dm = pkg.BatchDemod(64, bs, fs, fast_math=True)
        out = []
        for b in range(4):
            dm.set_chain(chain)
            assert dm.submit(blocks[b]) == 0
            dm.synchronize()
            out.append((dm.audio().reshape(64, -1).copy(), dm.rds_symbols()[1].copy()))
        dm.close()
        return out
    ref = short(False)
    for rep in range(8):
        got = short(True)
        for b in range(4):
            assert not np.isnan(got[b][0]).any(), (rep, b)
            assert rms(got[b][0].astype(np.float64) - ref[b][0]) <= 5e-6 and np.array_equal(got[b][1], ref[b][1]), (rep, b)


def test_one_launch_form_leaves_a_complete_state(pkg):
    """fmd_get_state / fmd_set_state across k_chain blocks: a station's state taken after four one-launch blocks and restored into another
    handle (other batch size, other block parity, three-launch form) continues bit-identically to what the FIRST handle produces when IT goes
    on in the three-launch form — i.e. the one-launch form keeps every history the other form reads (IQ tail, fm_out tail, the pilot points'
    last columns, the last span's cubic, the loop state, the L-R estimates) — through fmd_process_* (the caller's stream ordered behind the
    library's read) as well as fmd_submit_*."""
    import torch
    fs, bs, nb = 256_000, 16384, 9
    caps = _caps(9, nb * bs, float(fs), seed=9800)
    a = pkg.BatchDemod(9, bs, fs, fast_math=True)
    a.set_chain(True)
    for b in range(5):                                   # block 0: start-up (three launches); 1 .. 4 as k_chain, alternately submitted and processed
        t = torch.from_numpy(np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])).cuda()
        assert (a.process(t) if b % 2 else a.submit(t)) == 0
    a.synchronize()
    assert a.chain_blocks() == 4
    blob = a.get_state(7)
    a.set_chain(False)
    other = _caps(3, 2 * bs, float(fs), seed=9801)
    bdm = pkg.BatchDemod(3, bs, fs, fast_math=True)
    for b in range(2):
        bdm.process(other[:, b * bs:(b + 1) * bs])
    bdm.set_state(1, blob)
    for b in range(5, nb):
        a.process(caps[:, b * bs:(b + 1) * bs])
        x = np.ascontiguousarray(other[:, :bs]).copy()
        x[1] = caps[7, b * bs:(b + 1) * bs]
        bdm.process(x)
        assert np.array_equal(a.audio()[7].view(np.uint32), bdm.audio()[1].view(np.uint32)), b
        sa, ca = a.rds_symbols(); sb, cb = bdm.rds_symbols()
        assert ca[7] == cb[1] and np.array_equal(sa[7, :ca[7]].view(np.uint32), sb[1, :cb[1]].view(np.uint32)), b
    a.close(); bdm.close()


@pytest.mark.parametrize("n_ch", [6, 7])
def test_extract_stage_with_two_stations_per_workgroup_is_bit_identical(pkg, n_ch):
    """k_extract_bp<2> (round 6: tap tables and the block edge's matrix fetched once for two stations; the default from 3072 stations on) against
    one station per workgroup: every output bit for bit, over the block edge (S_old), acquisition and an odd station count (the last
    workgroup then has one station); stations with different cut-offs fall back to one per workgroup."""
    fs, bs, nb = 256_000, 16384, 5
    caps = _caps(n_ch, nb * bs, float(fs), seed=9900)
    outs = {}
    for mode in (1, 2):
        dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
        dm.set_extract_pairing(mode)
        au, sy = [], []
        for b in range(nb):
            assert dm.process(np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])) == 0
            au.append(dm.audio().copy()); s_, c_ = dm.rds_symbols(); sy.append((s_.copy(), c_.copy()))
            if b == 2: lm = dm.stream("lmr_est").copy()
        outs[mode] = (np.stack(au), sy, lm, dm.rds_bytes()[0].copy())
        dm.close()
    assert np.array_equal(outs[1][0].view(np.uint32), outs[2][0].view(np.uint32))
    assert np.array_equal(outs[1][2].view(np.uint32), outs[2][2].view(np.uint32)) and np.array_equal(outs[1][3], outs[2][3])
    for (sa, ca), (sb, cb) in zip(outs[1][1], outs[2][1]):
        assert np.array_equal(ca, cb) and all(np.array_equal(sa[c, :ca[c]].view(np.uint32), sb[c, :cb[c]].view(np.uint32)) for c in range(n_ch))
    # different cut-offs on one station: pairing is refused by the launcher (tables are per cut-off), outputs as without the request
    from fm_radio_amd.capi import default_controls
    dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
    dm.set_extract_pairing(1)
    ctl = default_controls(); ctl.lmr_cutoff_hz = 9000
    dm.set_controls(ctl, 1)
    ref = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
    ref.set_extract_pairing(2)
    ref.set_controls(ctl, 1)
    for b in range(2):
        blk = np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])
        assert dm.process(blk) == 0 and ref.process(blk) == 0
        assert np.array_equal(dm.audio().view(np.uint32), ref.audio().view(np.uint32))
    dm.close(); ref.close()
