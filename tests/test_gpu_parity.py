"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against the CPU oracle.

The oracle is handed the library's own designed coefficients (they equal the reference's except for the
pilot-peak gain, see test_coefficients_vs_oracle), and then every stream — discriminator/Hilbert output,
PLL phase, L+R, L-R, RDS baseband, RDS symbols, audio — is required to be BIT-IDENTICAL, RDS symbol
counts and Manchester bytes equal.  Against the golden fixtures dumped from the compiled reference the
north-star tolerance applies (audio / L-R / RDS symbols within 1e-4 RMS, RDS bytes identical).
"""
import hashlib

import numpy as np
import pytest

import oraclelib as O
import synth
from conftest import rms
from gpu_parity import compare_with_oracle, lib_coeffs_to_oracle, run_gpu
from rds_groups import decode_groups

pytestmark = pytest.mark.gpu

TOL_RMS = 1e-4  # BASELINE.json north_star


@pytest.fixture(scope="module")
def pkg():
    import fmradio_loader
    p = fmradio_loader.load()
    p.load_library()  # raises if the HIP extension was not built: no fallback
    import torch
    assert torch.cuda.is_available()
    return p


def _caps(n_ch, n, fs=1_024_000.0, seed=100, u8=False):
    conv = synth.to_u8 if u8 else synth.to_cf32
    return np.stack([conv(synth.fm_capture(n, fs=fs, seed=seed, channel=c)["iq"]) for c in range(n_ch)])


def _assert_exact(rep):
    bad = {k: rep["max_abs"].get(k) for k, v in rep["bit_exact"].items() if not v}
    assert not bad, f"streams not bit-identical to the oracle: {bad}"
    assert rep["rds_sym_equal_counts"] and rep["rds_bytes_equal"]


def test_single_channel_cf32(pkg):
    _assert_exact(compare_with_oracle(pkg, _caps(1, 6 * 8192), 8192, 1_024_000))


def test_multichannel_u8_ragged(pkg):
    # 5 channels: fewer than a wavefront, exercises the clamped lanes of the serial kernels
    _assert_exact(compare_with_oracle(pkg, _caps(5, 4 * 16384, u8=True), 16384, 1_024_000))


def test_block_65536_torch_device_pointer(pkg):
    _assert_exact(compare_with_oracle(pkg, _caps(2, 3 * 65536), 65536, 1_024_000, use_torch=True))


def test_blocks_longer_than_the_inline_lmr_phase_limit(pkg):
    """Long blocks (here 65536 samples at 256 kSa/s = 8192 audio samples, 820 L-R phase estimates per block)."""
    _assert_exact(compare_with_oracle(pkg, _caps(2, 3 * 65536, fs=256_000.0, seed=41), 65536, 256_000))


def test_more_than_one_wavefront_of_channels(pkg):
    # 70 channels -> two wavefronts, the second one mostly clamped
    caps = _caps(70, 2 * 4096, seed=300)
    rep = compare_with_oracle(pkg, caps, 4096, 1_024_000)
    _assert_exact(rep)


@pytest.mark.parametrize("fs,block", [(256_000, 4096), (2_048_000, 16384)])
def test_other_baseband_rates(pkg, fs, block):
    _assert_exact(compare_with_oracle(pkg, _caps(2, 4 * block, fs=float(fs)), block, fs))


def _ctl(pkg, **kw):
    from fm_radio_amd.capi import default_controls
    c = default_controls()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize("kw", [
    dict(use_deemphasis=1, deemphasis_tus=50),
    dict(use_deemphasis=1, deemphasis_tus=75, audio_out=1),
    dict(audio_out=0, lpr_cutoff_hz=12000),
    dict(audio_stereo_mix_factor=0.65, lmr_cutoff_hz=9000, lpr_cutoff_hz=100),
])
def test_controls(pkg, kw):
    _assert_exact(compare_with_oracle(pkg, _caps(2, 4 * 8192, seed=7), 8192, 1_024_000, controls=_ctl(pkg, **kw)))


def test_per_channel_controls(pkg):
    per = {0: _ctl(pkg, use_deemphasis=1, deemphasis_tus=50), 2: _ctl(pkg, audio_out=1, lmr_cutoff_hz=8000)}
    _assert_exact(compare_with_oracle(pkg, _caps(3, 4 * 8192, seed=9), 8192, 1_024_000, per_channel_controls=per))


def test_batched_equals_independent_runs(pkg):
    caps = _caps(4, 3 * 8192, seed=55)
    both = run_gpu(pkg, caps, 8192, 1_024_000)
    for c in range(4):
        one = run_gpu(pkg, caps[c:c + 1], 8192, 1_024_000)
        assert np.array_equal(both["audio"][c].view(np.uint32), one["audio"][0].view(np.uint32))
        assert np.array_equal(both["rds_sym"][c].view(np.uint32), one["rds_sym"][0].view(np.uint32))


@pytest.mark.parametrize("fs", [256_000, 1_024_000, 2_048_000])
@pytest.mark.parametrize("kw", [
    dict(),
    dict(use_deemphasis=1, deemphasis_tus=50, lpr_cutoff_hz=12000, lmr_cutoff_hz=9000),
    dict(use_deemphasis=1, deemphasis_tus=75, lpr_cutoff_hz=100, lmr_cutoff_hz=70000, audio_stereo_mix_factor=0.65),
])
def test_coefficients_vs_oracle(pkg, fs, kw):
    """The designer the LIBRARY runs (fmd_get_coeffs of a live handle) against the oracle's designs, at all three rates and with
    non-default cut-offs / de-emphasis (the parity comparisons hand the oracle the library's coefficients, so the
    designer has to be pinned on its own wherever those comparisons run).  The CPU twin is tests/test_capi_cpu.py::test_host_designer_matches_oracle."""
    dm = pkg.BatchDemod(1, 8192, fs)
    ctl = _ctl(pkg, **kw)
    if kw:
        dm.set_controls(ctl)
        n = 8192
        assert dm.process(np.zeros((1, n, 2), np.float32)) == 0       # controls take effect at a block boundary
    k = lib_coeffs_to_oracle(dm.get_coeffs(0))
    from gpu_parity import oracle_controls
    ref = O.design(fs, oracle_controls(ctl), rsqrt_mode=0)
    for name in ("b_fm_in", "b_fm_out", "b_hilbert", "pilot_a", "pll_lpf_b", "pll_lpf_a", "deemph_b", "deemph_a", "b_lpr", "b_lmr",
                 "b_rds", "ted_lpf_b", "ted_lpf_a", "bpsk_lpf_b", "bpsk_lpf_a", "pilot_b"):
        assert np.array_equal(k.arr(name).view(np.uint32), ref.arr(name).view(np.uint32)), (name, fs, kw)
    assert np.float32(k.fm_gain) == np.float32(ref.fm_gain)
    # vs the reference build's rsqrtss-approximated gain: a couple of ulp
    ref2 = O.design(fs, oracle_controls(ctl), rsqrt_mode=1)
    assert abs(float(k.arr("pilot_b")[0]) - float(ref2.arr("pilot_b")[0])) <= 4 * np.spacing(ref2.arr("pilot_b")[0])
    dm.close()


LOOP_TRACES = ["pilot", "pll", "pll_raw_err", "pll_pi_err", "bpsk_pll_sym", "bpsk_intdump", "bpsk_zcd", "bpsk_trig", "bpsk_ted_raw", "bpsk_ted_pi",
               "bpsk_pll_raw", "bpsk_pll_pi"]


@pytest.mark.parametrize("fs,block,n_ch,u8", [(1_024_000, 16384, 3, True), (256_000, 8192, 2, False), (1_024_000, 8192, 70, False)])
def test_loop_traces_bit_exact(pkg, fs, block, n_ch, u8):
    """The reference's per-sample loop getters — GetPilotOutput, GetPLLOutput, Get_PLL_Raw_Phase_Error_Output, Get_PLL_LPF_Phase_Error_Output
    (broadcast_fm_demod.h:245-248) and BPSK_Synchroniser's Get* views (bpsk_synchroniser.h:78-85) — through fmd_get_stream in the exact
    mode with FMD_FLAG_KEEP_TAPS: every one bit-identical to the oracle (which tests/test_oracle_vs_ref.py pins against the compiled
    reference), block after block from a cold start, i.e. through acquisition; and asking for them changes no output."""
    nb = 4
    caps = _caps(n_ch, nb * block, fs=float(fs), seed=17, u8=u8)
    dm = pkg.BatchDemod(n_ch, block, fs, keep_taps=True)
    plain = run_gpu(pkg, caps[: min(n_ch, 3)], block, fs)
    got = {k: [] for k in LOOP_TRACES + ["audio"]}
    for b in range(nb):
        assert dm.process(np.ascontiguousarray(caps[:, b * block:(b + 1) * block])) == 0
        for k in LOOP_TRACES:
            got[k].append(dm.stream(k))
        got["audio"].append(dm.audio().reshape(n_ch, -1))
    coeffs = [dm.get_coeffs(c) for c in range(n_ch)]
    dm.close()
    got = {k: np.concatenate(v, axis=1) for k, v in got.items()}
    for c in list(range(min(n_ch, 3))) + ([n_ch - 1] if n_ch > 3 else []):
        o = O.run_chain(caps[c], block, fs, u8=u8, coeffs=lib_coeffs_to_oracle(coeffs[c]), streams=LOOP_TRACES + ["audio"])
        for k in LOOP_TRACES:
            a, b_ = np.asarray(got[k][c], np.float32).reshape(-1), o[k].reshape(-1)
            assert a.shape == b_.shape, (k, a.shape, b_.shape)
            assert np.array_equal(a.view(np.uint32), b_.view(np.uint32)), (c, k, float(np.max(np.abs(a - b_))))
        assert np.array_equal(got["audio"][c].reshape(-1).view(np.uint32), o["audio"].reshape(-1).view(np.uint32))
    for c in range(min(n_ch, 3)):
        assert np.array_equal(got["audio"][c].view(np.uint32), plain["audio"][c].view(np.uint32))


def test_loop_traces_are_refused_where_they_do_not_exist(pkg):
    """Tolerance mode (loops at eight points per span / groups of four samples) and handles without FMD_FLAG_KEEP_TAPS: FMD_ERR_NAME."""
    for kw in (dict(keep_taps=True, fast_math=True), dict(keep_taps=False)):
        dm = pkg.BatchDemod(1, 8192, 1_024_000, **kw)
        assert dm.process(np.zeros((1, 8192, 2), np.float32)) == 0
        for k in ("pilot", "pll_pi_err", "bpsk_ted_pi"):
            with pytest.raises(Exception):
                dm.stream(k)
        dm.close()


def test_golden_chain_fixture(pkg, golden):
    """Against vectors dumped from the compiled reference (tests/golden/chain_b16384.npz)."""
    g = golden("chain_b16384.npz")
    out = run_gpu(pkg, g["capture"][None], 16384, 1_024_000)
    assert rms(out["audio"][0].reshape(-1) - g["audio"]) <= TOL_RMS
    assert rms(out["lmr"][0] - g["lmr"]) <= TOL_RMS
    assert rms(out["lpr"][0] - g["lpr"]) <= TOL_RMS
    assert rms(out["fm_out_iq"][0] - g["fm_out_iq"]) <= TOL_RMS
    assert np.array_equal(out["rds_count"][0], g["rds_count"])
    assert rms(out["rds_sym"][0] - g["rds_sym"]) <= TOL_RMS
    assert np.array_equal(out["rds_bytes"][0], g["rds_bytes"])


def test_long_run_rds_known_answer(pkg, golden):
    """2.6 s at block 65536 against the reference's dumped RDS stream; the synthesised groups must decode."""
    g = golden("long_b65536.npz")
    nb, bs, seed = int(g["n_blocks"]), int(g["block_size"]), int(g["seed"])
    cap = synth.to_u8(synth.fm_capture(nb * bs, seed=seed)["iq"])
    if hashlib.sha256(cap.tobytes()).hexdigest() != str(g["capture_sha256"]):
        pytest.skip("synthetic capture not bit-reproducible with this numpy build")
    out = run_gpu(pkg, cap[None], bs, 1_024_000)
    assert np.array_equal(out["rds_count"][0], g["rds_count"])
    assert np.array_equal(out["rds_bytes"][0], g["rds_bytes"])        # RDS bits: identical
    assert rms(out["rds_sym"][0] - g["rds_sym"]) <= TOL_RMS
    audio = out["audio"][0].reshape(nb, -1)
    for i, b in enumerate(g["audio_blocks"]):
        assert rms(audio[int(b)] - g["audio"][i]) <= TOL_RMS
    got = decode_groups(out["rds_bytes"][0])
    want = {tuple(int(v) for v in w) for w in g["groups"]}
    assert len(got) >= 20 and sum(1 for w in got if w in want) >= len(got) - 1


def test_wrong_size_block_is_dropped_and_reset(pkg):
    caps = _caps(1, 2 * 8192, seed=3)
    dm = pkg.BatchDemod(1, 8192, 1_024_000)
    assert dm.process(caps[:, :8192]) == 0
    a1 = dm.audio().copy()
    assert dm.process(caps[:, :4096]) == -2          # FMD_ERR_SIZE: dropped, nothing emitted
    assert np.array_equal(dm.audio(), a1)
    dm.reset()
    assert dm.process(caps[:, :8192]) == 0
    assert np.array_equal(dm.audio().view(np.uint32), a1.view(np.uint32))  # reset == fresh construction
    dm.close()


def test_roundtrip_properties_at_scale(pkg):
    """Full-size batch (4096 channels @ 256 kSa/s, BASELINE configs[2]) checked through size-independent properties:
    every channel equals the oracle-verified single-channel result of the capture it was given (tiled 8 distinct
    captures), and outputs are finite with the expected stereo content."""
    import torch
    n_ch, bs = 4096, 16384
    base = _caps(8, 2 * bs, fs=256_000.0, seed=900)
    small = run_gpu(pkg, base, bs, 256_000)
    dm = pkg.BatchDemod(n_ch, bs, 256_000)
    idx = np.arange(n_ch) % 8
    for b in range(2):
        blk = torch.from_numpy(np.ascontiguousarray(base[:, b * bs:(b + 1) * bs])).cuda()[torch.from_numpy(idx).cuda()].contiguous()
        assert dm.process(blk) == 0
    audio = dm.audio()
    ref_last = small["audio"][:, -audio.shape[1] * 2:].reshape(8, -1, 2)
    assert np.array_equal(audio.view(np.uint32), ref_last[idx].view(np.uint32))
    assert np.all(np.isfinite(audio))
    dm.close()


def test_device_atan2_equals_host_libm(pkg):
    """The device atan2f (FDLIBM restatement, branch-free) against the host libm the reference calls: bit-identical
    over random bit patterns, signal-range values, the range-reduction boundaries and every special value."""
    rng = np.random.default_rng(5)
    n = 8_000_000
    bits = rng.integers(0, 2**32, size=(2, n), dtype=np.uint64).astype(np.uint32)
    ys = [bits[0].view(np.float32), rng.uniform(-200, 200, n).astype(np.float32), rng.uniform(-1.5, 1.5, n).astype(np.float32)]
    xs = [bits[1].view(np.float32), rng.uniform(-200, 200, n).astype(np.float32), rng.uniform(-1.5, 1.5, n).astype(np.float32)]
    # ratios straddling the argument-reduction thresholds and the 2^+-60 / 2^25 shortcuts
    base = np.array([0.4375, 0.6875, 1.1875, 2.4375, 1.0, 2.0 ** -29, 2.0 ** 25, 2.0 ** 60, 2.0 ** -60, 2.0 ** 61, 2.0 ** -61], np.float32)
    r = np.concatenate([np.nextafter(b, np.float32(np.inf)) * (1 + np.arange(-2000, 2000, dtype=np.float32) * np.float32(6e-8)) for b in base]).astype(np.float32)
    xv = rng.choice(np.array([1.0, -1.0, 3.3, -0.77, 100.0, -2.0 ** -10], np.float32), r.size)
    ys.append((r * xv).astype(np.float32)); xs.append(xv)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e38, -1e38, 1e-38, 3e-39], np.float32)
    gy, gx = np.meshgrid(sp, sp)
    ys.append(gy.reshape(-1)); xs.append(gx.reshape(-1))
    y = np.concatenate(ys); x = np.concatenate(xs)
    host = O.libm_atan2f(y, x)
    for table_form in (False, True):   # the general form (serial loops, phase estimate) and the discriminator's table-driven form
        dev = pkg.selftest_atan2(y, x, table_form=table_form)
        nan = np.isnan(host) & np.isnan(dev)
        neq = (dev.view(np.uint32) != host.view(np.uint32)) & ~nan
        assert not neq.any(), (f"table_form={table_form}: {int(neq.sum())} mismatches, first: y={y[neq][0]!r} x={x[neq][0]!r} "
                               f"dev={dev[neq][0]!r} host={host[neq][0]!r}")


def test_device_atan2_on_u8_operands_is_libm_exhaustively(pkg):
    """The discriminator's arctangent on u8 IQ at 256 kSa/s (operands are the integers -127..128, zeros included, only 0/0
    treated specially): every one of the 65536 operand pairs against the host libm."""
    v = np.arange(-127, 129, dtype=np.float32)
    y, x = [a.reshape(-1).copy() for a in np.meshgrid(v, v)]
    dev = pkg.selftest_atan2(y, x, table_form="u8")
    host = O.libm_atan2f(y, x)
    neq = dev.view(np.uint32) != host.view(np.uint32)
    assert not neq.any(), f"{int(neq.sum())} mismatches, first: y={y[neq][0]!r} x={x[neq][0]!r} dev={dev[neq][0]!r} host={host[neq][0]!r}"


def test_device_atan2_short_form_is_exact_where_it_claims(pkg):
    """The locked-loop short form (no range selection, shortened division): wherever its predicate holds, its value is
    libm's atan2f bit-for-bit; and the predicate does hold on the inputs a locked loop produces."""
    rng = np.random.default_rng(11)
    n = 6_000_000
    bits = rng.integers(0, 2**32, size=(2, n), dtype=np.uint64).astype(np.uint32)
    # locked-loop shaped: x > 0, small |y/x|; then everything else, where the predicate must protect the value
    x0 = np.exp(rng.uniform(-9, 9, n)).astype(np.float32)     # inside the short form's [2^-14, 2^13.75) window
    y0 = (x0 * rng.uniform(-0.45, 0.45, n)).astype(np.float32)
    base = np.array([0.4375, 2.0 ** -29, 0.25, 1e-3], np.float32)
    r = np.concatenate([np.nextafter(b, np.float32(np.inf)) * (1 + np.arange(-3000, 3000, dtype=np.float32) * np.float32(6e-8)) for b in base]).astype(np.float32)
    xv = rng.choice(np.array([1.0, 3.3, 0.77, 100.0, 2.0 ** -10, 2.0 ** -14, 2.0 ** -15, 2.0 ** 13, 2.0 ** 14, 2.0 ** -62, 2.0 ** 61], np.float32), r.size)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e38, -1e38, 1e-38, 3e-39, 1e-20, 1e20], np.float32)
    gy, gx = np.meshgrid(sp, sp)
    y = np.concatenate([y0, bits[0].view(np.float32), rng.uniform(-1.5, 1.5, n).astype(np.float32), (r * xv * rng.choice(np.array([1, -1], np.float32), r.size)).astype(np.float32), gy.reshape(-1)])
    x = np.concatenate([x0, bits[1].view(np.float32), rng.uniform(-1.5, 1.5, n).astype(np.float32), xv, gx.reshape(-1)])
    dev, ok = pkg.selftest_atan2_small(y, x)
    host = O.libm_atan2f(y, x)
    neq = ok & (dev.view(np.uint32) != host.view(np.uint32))
    assert not neq.any(), f"{int(neq.sum())} mismatches, first: y={y[neq][0]!r} x={x[neq][0]!r} dev={dev[neq][0]!r} host={host[neq][0]!r}"
    with np.errstate(all="ignore"):
        t = np.abs(y[:n].astype(np.float64) / x[:n].astype(np.float64))
    inside = (t < 0.43) & (t > 1e-8)
    assert ok[:n][inside].mean() > 0.999          # the short division almost never needs its last refinement
    assert not ok[n:2 * n][x[n:2 * n] <= 0].any()  # never claimed outside x > 0


@pytest.mark.parametrize("pll_kernel", ["time_parallel", "time_parallel8", "low_work"])
def test_pilot_pll_out_of_its_comfort_zone(pkg, pll_kernel):
    """Stations the pilot PLL cannot hold: a pilot 130 Hz off (the NCO's +-100 Hz range saturates the control and the
    integrator: the clamp paths), no pilot at all (phase detector fed noise: every range of atan2f, the serial fall-back), a
    weak pilot under heavy noise — next to a normal station in the same wavefront.  Bit-identical to the oracle regardless."""
    n = 8 * 16384
    caps = np.stack([
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=501, channel=0)["iq"]),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=502, channel=1, pilot_hz=19130.0)["iq"]),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=503, channel=2, pilot_level=0.0)["iq"]),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=504, channel=3, pilot_level=0.02, noise_sigma=0.3)["iq"]),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=505, channel=4, pilot_hz=18870.0)["iq"]),
    ])
    _assert_exact(compare_with_oracle(pkg, caps, 16384, 256_000, pll_kernel=pll_kernel))


def test_block_length_that_only_fits_the_small_tiles(pkg):
    """5120-sample blocks at 256 kSa/s: 2560 fm_out samples (not a multiple of the 1024-sample front tile), 640 audio samples
    (the 128-sample extract tile), 20 PLL chunks — the less common kernel instantiations."""
    caps = _caps(3, 12 * 5120, fs=256_000.0, seed=311)
    _assert_exact(compare_with_oracle(pkg, caps, 5120, 256_000))


@pytest.mark.parametrize("pll_kernel", ["time_parallel", "time_parallel8", "low_work"])
def test_both_pilot_pll_kernels_are_bit_identical_to_the_oracle(pkg, pll_kernel):
    """The library picks the pilot-PLL kernel by batch size (time-parallel up to 8192 channels, low-work above); both must
    produce the oracle's bits, through acquisition and in lock, for channel counts that do not fill a wavefront."""
    caps = _caps(5, 10 * 16384, fs=256_000.0, seed=77)
    _assert_exact(compare_with_oracle(pkg, caps, 16384, 256_000, pll_kernel=pll_kernel))


def test_pll_speculation_commits_long_spans_in_lock_and_short_ones_before(pkg):
    """The pilot PLL kernel evaluates 16 samples at a time under the assumption that the NCO frequency word stays put and
    commits the prefix for which that held.  Results are identical either way (every other test in this file checks that) —
    here only that the statistics look as designed: short spans while acquiring, ~15 of 16 samples per span in lock."""
    nb = 14
    caps = _caps(2, nb * 16384, fs=256_000.0, seed=23)
    dm = pkg.BatchDemod(n_channels=2, block_size=16384, fs_baseband=256_000)
    per_block = []
    for b in range(nb):
        dm.process(caps[:, b * 16384:(b + 1) * 16384])
        per_block.append(dm.spec_stats(reset=True)["pll"])
    dm.close()
    assert per_block[0]["chunks"] == 8192 // 128
    assert per_block[-1]["samples"] == 2 * 8192                     # every sample of both channels went through a span
    assert per_block[0]["samples_per_span"] < 12.0 or per_block[0]["sequence_spans"] > 0   # acquisition: the frequency word moves all the time
    locked = per_block[-4:]
    assert all(p["samples_per_span"] > 13.5 for p in locked), locked  # in lock: it changes on ~0.5 % of the samples
    assert sum(p["serial_chunks"] for p in locked) == 0


@pytest.mark.parametrize("pll_kernel,k", [("time_parallel", 16), ("time_parallel8", 8)])
def test_loops_out_of_lock_run_the_sequence_form_not_the_serial_iteration(pkg, pll_kernel, k):
    """Round 6: a station that cannot hold lock moves its NCO frequency word on every sample, so "the word stays put" commits
    one sample a span; its wavefront then speculates on the SEQUENCE of words (a guess pass, then the exact pass confirms word by word) and
    commits whole spans again.  From the second block on (the first finds out in its first chunk and hands the rest to the other loop): no serial
    chunks anywhere, the sequence form on every span, nearly K samples per span — next to a normal station in the same wavefront, and bit-identical to the
    oracle (the serial restatement of the reference)."""
    nb, bs = 6, 16384
    rows = [synth.to_cf32(synth.fm_capture(nb * bs, fs=256_000.0, seed=900 + c, channel=c, **kw)["iq"])
            for c, kw in enumerate([{}, {"pilot_level": 0.0}, {"pilot_hz": 19130.0}])]
    caps = np.stack(rows)
    _assert_exact(compare_with_oracle(pkg, caps, bs, 256_000, pll_kernel=pll_kernel))
    dm = pkg.BatchDemod(n_channels=3, block_size=bs, fs_baseband=256_000, pll_kernel=pll_kernel)
    per_block = []
    for b in range(nb):
        dm.process(caps[:, b * bs:(b + 1) * bs])
        per_block.append(dm.spec_stats(reset=True)["pll"])
    dm.close()
    assert per_block[0]["serial_chunks"] == 0 and per_block[0]["sequence_spans"] > 0        # the block that finds out: one slow chunk, then the sequence form
    for p in per_block[1:]:
        assert p["serial_chunks"] == 0, per_block
        assert p["sequence_spans"] >= 0.95 * (bs // 2) / k, per_block                       # (one wavefront: its spans are the slowest station's)
        assert p["samples_per_span"] > 0.9 * k, per_block


@pytest.mark.parametrize("thresholds,calm,busy", [((4, 7168), 8, 16), ((4, 4), "low-work", 16)])
def test_pilot_pll_kernel_follows_what_is_out_of_lock(pkg, thresholds, calm, busy):
    """Between 3585 and 4096 stations the exact mode runs the pilot-PLL kernel with 8 lanes a station while every loop holds lock and with 16
    while some do not (a loop out of lock costs its wavefront ~1.4x and the kernel lasts as long as its slowest wavefront); above 7168 stations
    the low-work kernel gives way to the time-parallel one (8 lanes; 16 up to 4096 stations), whose sequence form gets through such loops.  The
    test hook moves those ranges down to this batch: one station's pilot is 130 Hz off for ten blocks in the middle of the capture.  Every
    stream bit-identical to the oracle through both switches, and the statistics show the calm kernel, then the busy one, then the calm one."""
    bs, C = 16384, 12
    parts = (14, 10, 18)
    def cap(c, nblk, **kw):
        return synth.to_cf32(synth.fm_capture(nblk * bs, fs=256_000.0, seed=1300 + c, channel=c, **kw)["iq"])
    rows = [cap(c, sum(parts)) for c in range(C)]
    rows[5] = np.concatenate([cap(5, parts[0]), cap(50, parts[1], pilot_hz=19130.0), cap(51, parts[2])])
    caps = np.stack(rows)
    _assert_exact(compare_with_oracle(pkg, caps, bs, 256_000, pll_k16_max=thresholds))
    dm = pkg.BatchDemod(n_channels=C, block_size=bs, fs_baseband=256_000)
    dm.pll_adaptive(*thresholds)
    seen = []
    for b in range(sum(parts)):
        dm.process(caps[:, b * bs:(b + 1) * bs])
        dm.synchronize()
        p = dm.spec_stats(reset=True)["pll"]
        # (the low-work kernel counts 16-sample chunks and no spans; the time-parallel kernel 128-sample chunks per wavefront of 64 / lanes stations)
        seen.append("low-work" if p["spans"] == 0 else (16 if p["samples_per_span"] > 8.0 else 8))
    dm.close()
    a, b = parts[0], parts[0] + parts[1]
    assert seen[a - 2:a] == [calm, calm], seen                                      # in lock, the counter at rest for 8 launches
    assert seen[b - 2:b] == [busy, busy], seen                                      # the pilot off: within a few blocks (the counter's way back to the host)
    assert seen[-2:] == [calm, calm], seen                                          # back in lock


@pytest.mark.parametrize("thresholds,pipelined", [(None, True), ((2, 7168), True), ((2, 2), True), ((2, 2), False)])
def test_loops_that_wander_in_and_out_of_lock(pkg, thresholds, pipelined):
    """Stations whose pilot loop neither holds lock nor loses it for good — receiver noise only, a pilot at the noise floor, a dead front end
    whose zeros carry signs — next to a normal station, sixteen blocks: the wavefront changes between the constant-word and the sequence form
    chunk by chunk, and (with the test hook's thresholds) the library between the 8- and 16-lane and the low-work kernels block by block.
    Whatever ran, every stream is the oracle's bit for bit."""
    n, bs = 16 * 16384, 16384
    rng = np.random.default_rng(77)
    caps = np.stack([
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=601, channel=0)["iq"]),
        (0.02 * rng.standard_normal((n, 2))).astype(np.float32),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=603, channel=2, pilot_level=0.02, noise_sigma=0.3)["iq"]),
        np.where(rng.random((n, 2)) < 0.5, np.float32(0.0), np.float32(-0.0)).astype(np.float32),
        synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=605, channel=4, pilot_level=0.01, noise_sigma=0.1)["iq"]),
    ])
    _assert_exact(compare_with_oracle(pkg, caps, bs, 256_000, pll_k16_max=thresholds, pipelined=pipelined))      # (not pipelined: FMD_FLAG_NO_PIPELINE, every stage on the caller's stream)


@pytest.mark.parametrize("fs,u8", [(1_024_000, True), (2_048_000, False)])
def test_loops_out_of_lock_behind_the_first_decimator(pkg, fs, u8):
    """The same at the reference's own rate and capture format (u8 at 1.024 MSa/s) and at 2.048 MSa/s: a station without pilot and one whose pilot is
    130 Hz off, next to a normal one; the sequence form from the second block on, bit-identical."""
    bs = fs * 64 // 1000
    conv = synth.to_u8 if u8 else synth.to_cf32
    caps = np.stack([conv(synth.fm_capture(6 * bs, fs=float(fs), seed=950 + c, channel=c, **kw)["iq"])
                     for c, kw in enumerate([{}, {"pilot_level": 0.0}, {"pilot_hz": 19130.0}])])
    _assert_exact(compare_with_oracle(pkg, caps, bs, fs))


def test_blocks_with_more_symbols_than_the_sign_buffer_holds(pkg):
    """k_rds_sync buffers the symbols' signs for the Manchester decoder in LDS, 1024 per station; a block of
    131072 samples at 256 kSa/s (8192 RDS samples, ~1200 symbols; ~1790 at the symbol clock's upper rail) overflows one buffer, so the
    decoder runs whenever a row is nearly full.  Symbols, counts and bytes bit-identical to the oracle, both ingest paths; and the
    tolerance mode (single-pass synchroniser) carries the same bits."""
    bs, fs = 131072, 256_000
    caps = _caps(3, 3 * bs, fs=float(fs), seed=8800)
    rep = compare_with_oracle(pkg, caps, bs, fs)
    _assert_exact(rep)
    g = run_gpu(pkg, caps[:1], bs, fs)
    assert int(g["rds_count"][0].max()) > 1055            # the case: more symbols in a block than one buffer row holds
    import test_gpu_fast as F
    import oraclelib as O
    f = run_gpu(pkg, caps, bs, fs, fast_math=True)
    for c in range(3):
        o = O.run_chain(caps[c], bs, fs, u8=False, coeffs=lib_coeffs_to_oracle(f["coeffs"][c]), streams=["rds_sym"])
        assert F.same_bits_once_in_lock(f["rds_bytes"][c], o["rds_bytes"], skip_bits=5 * 76), c


def test_pcm16_audio_frames_match_the_scraper_conversion(pkg):
    """fmd_audio_pcm16_dev (the multi-GPU gather's payload): the reference scraper's float -> int16 conversion of the audio
    block (fm_scraper.cpp:79-82: sample * (32767 * 0.95f), truncated toward zero), on the device, behind the block's outputs."""
    import torch
    caps = _caps(3, 4 * 16384, fs=256_000.0, seed=77)
    dm = pkg.BatchDemod(3, 16384, 256_000)
    out = torch.empty((3, dm.rates.n_audio, 2), dtype=torch.int16, device="cuda")
    side = torch.cuda.Stream()
    for b in range(4):
        dm.process(caps[:, b * 16384:(b + 1) * 16384])
        dm.audio_pcm16_into(out, side)
        side.synchronize()
        want = (dm.audio() * (np.float32(32767.0) * np.float32(0.95))).astype(np.int32).astype(np.int16)
        assert np.array_equal(out.cpu().numpy(), want), b
        assert np.abs(want).max() > 1000      # a real signal, not silence
    dm.close()


def test_pll_handover_per_wavefront_equals_stream_order(pkg):
    """Batches up to 2816 stations hand the PLL state from one block's k_pilot_pll launch to the next per wavefront, while both
    launches are resident (two streams, release/acquire on a per-wavefront sequence number) — the same bits as ordering the two
    launches by the stream (FMD_FLAG_PLL_STREAM_ORDER), block after block, for a batch that spans many wavefronts."""
    import torch
    n_ch, bs, nb = 600, 16384, 10
    base = _caps(5, nb * bs, fs=256_000.0, seed=2100)
    caps = torch.from_numpy(np.ascontiguousarray(base[np.arange(n_ch) % 5])).cuda()
    outs = []
    for stream_order in (False, True):
        dm = pkg.BatchDemod(n_ch, bs, 256_000, keep_taps=True, pll_stream_order=stream_order)
        got = []
        for b in range(nb):
            dm.process(caps[:, b * bs:(b + 1) * bs].contiguous())
            if b % 3 == 2 or b == nb - 1:      # not after every block: let several launches be in flight
                got.append((dm.stream("pll_dt").copy(), dm.audio().copy()))
        dm.close()
        outs.append(got)
    for (dt_a, au_a), (dt_b, au_b) in zip(*outs):
        assert np.array_equal(dt_a.view(np.uint32), dt_b.view(np.uint32))
        assert np.array_equal(au_a.view(np.uint32), au_b.view(np.uint32))


def test_two_demodulators_interleaved_and_reset_midstream(pkg):
    """Two handles advanced alternately (each with its own per-wavefront PLL hand-over chain and streams), one of them reset
    half way: every block equals what a lone, sequentially executed demodulator produces for the same input history."""
    n_ch, bs, nb = 96, 16384, 6
    caps = _caps(4, nb * bs, fs=256_000.0, seed=909)
    caps = np.ascontiguousarray(caps[np.arange(n_ch) % 4])
    def lone(reset_at):
        dm = pkg.BatchDemod(n_ch, bs, 256_000, pipelined=False)
        out = []
        for b in range(nb):
            if b == reset_at:
                dm.reset()
            dm.process(caps[:, b * bs:(b + 1) * bs])
            out.append((dm.audio().copy(), dm.stream("pll_dt").copy()))
        dm.close()
        return out
    want_a, want_b = lone(None), lone(3)
    a = pkg.BatchDemod(n_ch, bs, 256_000)
    b_ = pkg.BatchDemod(n_ch, bs, 256_000)
    for blk in range(nb):
        if blk == 3:
            b_.reset()
        a.process(caps[:, blk * bs:(blk + 1) * bs])
        b_.process(caps[:, blk * bs:(blk + 1) * bs])
        for dm, want in ((a, want_a), (b_, want_b)):
            assert np.array_equal(dm.audio().view(np.uint32), want[blk][0].view(np.uint32)), blk
            assert np.array_equal(dm.stream("pll_dt").view(np.uint32), want[blk][1].view(np.uint32)), blk
    a.close(); b_.close()


def test_gpu_runs_are_deterministic(pkg):
    caps = _caps(3, 6 * 65536, seed=17, u8=True)
    a = run_gpu(pkg, caps, 65536, 1_024_000)
    b = run_gpu(pkg, caps, 65536, 1_024_000)
    for k in ("audio", "fm_out_iq", "pll_dt", "rds"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert np.array_equal(a["rds_count"], b["rds_count"])


def test_pipelined_and_sequential_execution_agree_at_scale(pkg):
    """1500 stations x 8 blocks: the five-stream pipeline with four blocks in flight against every stage back to back on one
    stream — any missing dependency between stages of neighbouring blocks would show up as a difference (run twice: a race
    need not fire every time)."""
    import torch
    n_ch, bs, nb = 1500, 16384, 8
    base = _caps(6, nb * bs, fs=256_000.0, seed=1300)
    idx = torch.from_numpy(np.arange(n_ch) % 6).cuda()
    blocks = [torch.from_numpy(np.ascontiguousarray(base[:, b * bs:(b + 1) * bs])).cuda()[idx].contiguous() for b in range(nb)]

    def run(pipelined):
        dm = pkg.BatchDemod(n_ch, bs, 256_000, pipelined=pipelined)
        for blk in blocks[:-1]:
            assert dm.process(blk) == 0                      # queued back to back: up to four blocks in flight, no host sync
        assert dm.process(blocks[-1]) == 0
        a = dm.audio()                                       # every stage's state feeds forward: one wrong sample anywhere shows here
        by, cnt = dm.rds_bytes()
        syms, scnt = dm.rds_symbols()
        dm.close()
        return (hashlib.sha256(a.tobytes()).hexdigest(), hashlib.sha256(by.tobytes() + cnt.tobytes()).hexdigest(),
                hashlib.sha256(syms.tobytes() + scnt.tobytes()).hexdigest())

    ref = run(False)
    assert run(True) == ref
    assert run(True) == ref


def test_cpp_host_adaptor_matches_oracle(pkg, tmp_path):
    """The C++ adaptor (fm-radio_amd/host/broadcast_fm_demod_gpu.hpp: reference method names over the C ABI, App-style u8
    re-blocking) driven like the reference's own mains, against the oracle."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "adaptor_main"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{root / 'include'}", f"-I{root / 'fm-radio_amd' / 'host'}", str(root / "tests" / "cpp" / "adaptor_main.cpp"),
                    f"-L{root / 'fm-radio_amd' / 'csrc'}", "-lfmdemod", f"-Wl,-rpath,{root / 'fm-radio_amd' / 'csrc'}", "-o", str(exe)], check=True)
    bs, nb = 16384, 8
    cap = synth.to_u8(synth.fm_capture(bs * nb + 777, seed=31)["iq"])   # trailing partial block is never processed
    cap.tofile(tmp_path / "cap.u8")
    subprocess.run([str(exe), str(tmp_path / "cap.u8"), str(tmp_path), str(bs)], check=True)
    dm = pkg.BatchDemod(1, bs, 1_024_000)
    k = lib_coeffs_to_oracle(dm.get_coeffs(0))
    dm.close()
    o = O.run_chain(cap[: bs * nb], bs, 1_024_000, u8=True, coeffs=k, streams=["audio", "rds_sym", "lpr", "pll", "pll_pi_err", "pilot", "bpsk_ted_pi", "bpsk_pll_sym"])
    for name, key in (("audio.f32", "audio"), ("rds_sym.f32", "rds_sym"), ("lpr.f32", "lpr"), ("pll.cf32", "pll"), ("pll_pi_err.f32", "pll_pi_err"),
                      ("pilot.cf32", "pilot"), ("bpsk_ted_pi.f32", "bpsk_ted_pi"), ("bpsk_pll_sym.cf32", "bpsk_pll_sym")):
        got = np.fromfile(tmp_path / name, dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), o[key].view(np.uint32)), name
    assert np.array_equal(np.fromfile(tmp_path / "rds_bytes.u8", dtype=np.uint8), o["rds_bytes"])
