"""GPU tests (pytest -m gpu) of the library at the sizes and in the regime bench.py measures: thousands of stations, many
blocks submitted back to back with no host synchronisation in between (all pipeline stages and buffer slots in use, loops
in lock), default kernel selection.  Every channel is compared with an ORACLE-VERIFIED small run of the capture it was given
(8 distinct captures tiled over the batch), on audio, PLL phase, RDS symbols and Manchester bytes.

Also here: the output-lifetime contract (FMD_OUTPUT_LIFETIME_BLOCKS, fmd_release_outputs), the per-channel state snapshot,
the low-work PLL kernel in the pipelined regime, and the 2-rank plumbing of `bench.py --gpus 2` as the driver invokes it.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import synth
import oraclelib as O
from gpu_parity import compare_with_oracle, lib_coeffs_to_oracle, run_gpu

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


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


def _assert_exact(rep):
    bad = {k: rep["max_abs"].get(k) for k, v in rep["bit_exact"].items() if not v}
    assert not bad, f"streams not bit-identical to the oracle: {bad}"
    assert rep["rds_sym_equal_counts"] and rep["rds_bytes_equal"]


def _tiled_run(pkg, base, n_ch, bs, fs, nb, read_at, **demod_kw):
    """Feed base[c % len(base)] to channel c of an n_ch batch, nb blocks back to back; host-read the outputs only after the
    blocks listed in read_at.  Returns {block: dict of outputs}."""
    import torch
    idx = torch.from_numpy(np.arange(n_ch) % base.shape[0]).cuda()
    dbase = torch.from_numpy(base).cuda()
    dm = pkg.BatchDemod(n_ch, bs, fs, **demod_kw)
    got = {}
    for b in range(nb):
        blk = dbase[:, b * bs:(b + 1) * bs][idx].contiguous()
        assert dm.process(blk) == 0
        if b in read_at:
            syms, cnt = dm.rds_symbols()
            by, bc = dm.rds_bytes()
            # the PLL's result: the per-sample phase stream, or (tolerance mode, which does not materialise it) the span polynomials
            got[b] = dict(audio=dm.audio(), pll_dt=dm.stream("pll_poly" if demod_kw.get("fast_math") else "pll_dt"), syms=syms, cnt=cnt, by=by, bc=bc)
    spec = dm.spec_stats()
    dm.close()
    return got, spec


def _check_against_small(got, small, n_base, n_ch, bs_audio, n_fm_out, pll="pll_dt"):
    """small: run_gpu() result of the n_base base captures (every block concatenated)."""
    idx = np.arange(n_ch) % n_base
    for b, g in got.items():
        want_audio = small["audio"][:, b * 2 * bs_audio:(b + 1) * 2 * bs_audio].reshape(n_base, bs_audio, 2)
        assert np.array_equal(g["audio"].view(np.uint32), want_audio[idx].view(np.uint32)), f"audio, block {b}"
        w = n_fm_out if pll == "pll_dt" else 4 * (1 + n_fm_out // 128)
        want_dt = small[pll][:, b * w:(b + 1) * w]
        assert np.array_equal(g["pll_dt"].view(np.uint32), want_dt[idx].view(np.uint32)), f"pll_dt, block {b}"
        want_cnt = small["rds_count"][:, b]
        assert np.array_equal(g["cnt"], want_cnt[idx]), f"rds symbol counts, block {b}"
        for c in range(0, n_ch, 97):      # symbol values and bytes on a sample of channels (rows are ragged)
            k = idx[c]
            lo = int(small["rds_count"][k, :b].sum())
            assert np.array_equal(g["syms"][c, :g["cnt"][c]].view(np.uint32), small["rds_sym"][k][lo:lo + want_cnt[k]].view(np.uint32)), (b, c)


@pytest.mark.parametrize("fs,u8,n_ch", [(256_000, False, 4096), (256_000, True, 4096), (1_024_000, False, 4096), (256_000, False, 8192)])
def test_bench_path_in_lock_against_oracle_verified_tiles(pkg, fs, u8, n_ch):
    """The exact path bench.py times — BASELINE configs[2] (4096 stations @ 256 kSa/s, cf32 and u8), the reference's native
    1.024 MSa/s rate (k_predecim on its own stream), and the per-GPU shard of configs[3] (8192 stations: low-work PLL kernel
    k_pilot_pll_pairs, sample-by-sample k_pilot_power) — 14 blocks submitted back to back with the default kernel selection,
    six buffer slots and all streams in use, loops in lock from block ~8 on.  Anchored to the oracle through 8 base captures."""
    nb = 14
    bs = fs * 64 // 1000
    base = _caps(8, nb * bs, float(fs), seed=4100 + (1 if u8 else 0), u8=u8)
    _assert_exact(compare_with_oracle(pkg, base, bs, fs))                 # the tiles themselves equal the oracle, every stream
    small = run_gpu(pkg, base, bs, fs)
    got, spec = _tiled_run(pkg, base, n_ch, bs, fs, nb, read_at={4, 9, 11, 13})
    n_fm_out = bs // (fs // 256_000) // 2
    _check_against_small(got, small, 8, n_ch, n_fm_out // 4, n_fm_out)
    assert spec["pll"]["chunks"] > 0


@pytest.mark.parametrize("fs,u8,n_ch", [(256_000, False, 4096), (256_000, True, 4096), (1_024_000, False, 4096), (256_000, False, 8192)])
def test_tolerance_mode_bench_path_in_lock_against_oracle_verified_tiles(pkg, fs, u8, n_ch):
    """The path bench.py times by default (FMD_FLAG_FAST_MATH): the same four configurations, 14 blocks back to back.  The tolerance
    mode is deterministic and independent of the batch a station is in, so the large pipelined run must reproduce, bit for bit, the
    small run of the 8 base captures — and that small run is compared with the oracle under the north-star tolerance."""
    import test_gpu_fast as F
    nb = 14
    bs = fs * 64 // 1000
    base = _caps(8, nb * bs, float(fs), seed=4200 + (1 if u8 else 0), u8=u8)
    small = run_gpu(pkg, base, bs, fs, fast_math=True)
    for c in range(8):
        o = O.run_chain(base[c], bs, fs, u8=u8, coeffs=lib_coeffs_to_oracle(small["coeffs"][c]), streams=["fm_out_iq", "lpr", "lmr", "audio", "rds_sym", "lmr_phase"])
        for k in ("fm_out_iq", "lpr"):
            assert F.rms(np.asarray(small[k][c], np.float64).reshape(-1) - o[k].reshape(-1)) <= F.TOL_RMS, (c, k)
        ex, _, _, _ = F.lmr_audio_excess(small, o, c, nb)      # L-R and audio: every block within 1e-4 (the helper documents the one allowance)
        assert ex <= 1.0, (c, ex)
        assert F.same_bits_once_in_lock(small["rds_bytes"][c], o["rds_bytes"], skip_bits=5 * 76), c
    got, spec = _tiled_run(pkg, base, n_ch, bs, fs, nb, read_at={4, 9, 11, 13}, fast_math=True)
    n_fm_out = bs // (fs // 256_000) // 2
    _check_against_small(got, small, 8, n_ch, n_fm_out // 4, n_fm_out, pll="pll_poly")
    assert spec["pll"]["samples_per_span"] == 128.0


def test_low_work_pll_kernel_pipelined_without_host_sync(pkg):
    """FMD_FLAG_PLL_LOW_WORK on a small batch in pipelined mode must keep consecutive blocks' k_pilot_pll_pairs
    launches ordered (the kernel has no per-wavefront hand-over) — 12 blocks back to back, no per-block sync, against the
    sequential execution; repeated, a race need not fire every time."""
    n_ch, bs, nb = 700, 16384, 12
    base = _caps(5, nb * bs, 256_000.0, seed=5200)
    want, _ = _tiled_run(pkg, base, n_ch, bs, 256_000, nb, read_at={nb - 1}, pipelined=False, pll_kernel="low_work")
    for _ in range(3):
        got, _ = _tiled_run(pkg, base, n_ch, bs, 256_000, nb, read_at={nb - 1}, pll_kernel="low_work")
        for k in ("audio", "pll_dt", "cnt", "by"):
            assert np.array_equal(got[nb - 1][k], want[nb - 1][k]), k


@pytest.mark.parametrize("fs,u8,fast", [(256_000, False, True), (256_000, True, True), (1_024_000, False, True), (1_024_000, True, True),
                                        (256_000, False, False), (1_024_000, True, False)])
def test_device_input_at_any_sample_alignment(pkg, fs, u8, fast):
    """The reference takes a span of samples at whatever address (broadcast_fm_demod.h:231); the kernels read 16 bytes per lane where the
    capture allows.  A device block that starts one sample (8 / 2 bytes) into an allocation gives the same bits as an aligned one."""
    import torch
    bs, n_ch, nb = fs * 64 // 1000, 3, 3
    caps = _caps(n_ch, nb * bs, float(fs), seed=7700, u8=u8)
    outs = []
    for shift in (0, 1):
        dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=fast)
        audio = []
        for b in range(nb):
            blk = torch.from_numpy(np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])).cuda()
            if shift:
                flat = torch.empty(blk.numel() + 2 * shift, dtype=blk.dtype, device="cuda")
                view = flat[2 * shift:].view(n_ch, bs, 2)
                view.copy_(blk)
                blk = view
                assert blk.data_ptr() % 16 != 0 and blk.is_contiguous()
            assert dm.process(blk) == 0
            audio.append(dm.audio().copy())
        by, bc = dm.rds_bytes()
        outs.append((np.concatenate(audio, axis=1), by.copy(), bc.copy()))
        dm.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_outputs_stay_valid_for_the_stated_number_of_blocks(pkg):
    """include/fmdemod.h: a block's output views stay valid while at most FMD_OUTPUT_LIFETIME_BLOCKS further blocks have been
    submitted.  Read block 3's device views after exactly that many more submissions (no host sync in between)."""
    import torch
    from fm_radio_amd import capi
    life = capi.FMD_OUTPUT_LIFETIME_BLOCKS
    assert pkg.load_library().fmd_output_lifetime_blocks() == life
    n_ch, bs = 1024, 16384
    nb = 4 + life
    base = _caps(4, nb * bs, 256_000.0, seed=6100)
    idx = torch.from_numpy(np.arange(n_ch) % 4).cuda()
    dbase = torch.from_numpy(base).cuda()
    ref = pkg.BatchDemod(n_ch, bs, 256_000, pipelined=False)
    for b in range(4):
        ref.process(dbase[:, b * bs:(b + 1) * bs][idx].contiguous())
    want = ref.audio()
    ref.close()
    dm = pkg.BatchDemod(n_ch, bs, 256_000)
    view = None
    for b in range(nb):
        dm.process(dbase[:, b * bs:(b + 1) * bs][idx].contiguous())
        if b == 3:
            view = dm.audio_tensor()          # device view of block 3's audio
    dm.synchronize()
    assert np.array_equal(view.cpu().numpy().view(np.uint32), want.view(np.uint32))
    dm.close()


@pytest.mark.parametrize("fast,n_ch,submit", [(False, 512, False), (True, 512, False), (True, 2048, False), (True, 2048, True)])
def test_release_outputs_holds_a_slot_for_a_slow_consumer(pkg, fast, n_ch, submit):
    """fmd_release_outputs: a consumer that is still reading block b's audio view when b + 6 is submitted (the call that
    reuses the slot) is waited for on the device.  The consumer stream is stalled with a long spin kernel before its copy.
    Both modes; in the tolerance mode below and above the batch size of the deferred schedule, through fmd_process_* and fmd_submit_*:
    a stalled extract stage must also hold back the pilot stage that writes the history in front of its slot's rows."""
    import torch
    bs, nb = 16384, 16
    base = _caps(4, nb * bs, 256_000.0, seed=6200)
    idx = torch.from_numpy(np.arange(n_ch) % 4).cuda()
    dbase = torch.from_numpy(base).cuda()
    blocks = [dbase[:, b * bs:(b + 1) * bs][idx].contiguous() for b in range(nb)]
    ref = pkg.BatchDemod(n_ch, bs, 256_000, pipelined=False, fast_math=fast)
    want = []
    for b in range(nb):
        ref.process(blocks[b])
        want.append(ref.audio().copy())
    ref.close()
    dm = pkg.BatchDemod(n_ch, bs, 256_000, fast_math=fast)
    side = torch.cuda.Stream()
    kept = {}
    for b in range(nb):
        (dm.submit if submit else dm.process)(blocks[b])
        if b in (2, 5):
            view = dm.audio_tensor()
            with torch.cuda.stream(side):
                dm.wait_outputs(side)
                torch.cuda._sleep(int(2.0e8))             # ~0.1 s: far longer than the 13 remaining blocks take
                kept[b] = view.clone()
                dm.release_outputs(side)
    dm.synchronize()
    side.synchronize()
    for b, t in kept.items():
        assert np.array_equal(t.cpu().numpy().view(np.uint32), want[b].view(np.uint32)), b
    assert np.array_equal(dm.audio().view(np.uint32), want[-1].view(np.uint32))
    dm.close()


@pytest.mark.parametrize("fs,n_ch,bs,u8", [(256_000, 3072, 16384, False), (1_024_000, 1024, 65536, False), (1_024_000, 1024, 65536, True)])
def test_submit_schedule_and_output_lag_in_the_tolerance_mode(pkg, fs, n_ch, bs, u8):
    """(1.024 MSa/s: the schedule with the front end — k_front_pre_mfma, the previous block's pilot stage riding it — on the capture's
    queue and the extract stages on the front end's.)
    fmd_submit_*_dev on a batch large enough for the deferred schedule (the extract stage of block k queued behind the front end of
    block k + 1, include/fmdemod.h fmd_set_output_lag): every block's outputs equal those of the block-by-block run, whichever way the
    caller asks for them —
      default: wait_outputs right after submit(k) gives block k (and switches the handle to queue-at-submission);
      a free-running caller gets the last block after synchronize();
      fmd_set_output_lag(1): the device views after submit(k) are block k - 1's, block k's after synchronize()."""
    import torch
    nb = 9
    base = _caps(4, nb * bs, float(fs), seed=6300, u8=u8)
    idx = torch.from_numpy(np.arange(n_ch) % 4).cuda()
    dbase = torch.from_numpy(base).cuda()
    blocks = [dbase[:, b * bs:(b + 1) * bs][idx].contiguous() for b in range(nb)]
    ref = pkg.BatchDemod(n_ch, bs, fs, pipelined=False, fast_math=True)
    want, want_bytes = [], []
    for b in range(nb):
        ref.process(blocks[b])
        want.append(ref.audio().copy())
        want_bytes.append(ref.rds_bytes())
    ref.close()
    side = torch.cuda.Stream()

    def same(t, b):
        return np.array_equal(t.cpu().numpy().view(np.uint32), want[b].view(np.uint32))

    # free-running: nothing asked for until the end
    dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
    for b in range(nb):
        dm.submit(blocks[b])
    dm.synchronize()
    assert np.array_equal(dm.audio().view(np.uint32), want[-1].view(np.uint32))
    by, cnt = dm.rds_bytes()
    assert np.array_equal(cnt, want_bytes[-1][1]) and np.array_equal(by, want_bytes[-1][0])
    dm.close()
    # default: the newest block's outputs on request
    dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
    kept = {}
    for b in range(nb):
        dm.submit(blocks[b])
        if b in (0, 3, 4, nb - 1):
            with torch.cuda.stream(side):
                dm.wait_outputs(side)
                kept[b] = dm.audio_tensor().clone()
                dm.release_outputs(side)
    dm.synchronize(); side.synchronize()
    for b, t in kept.items():
        assert same(t, b), b
    dm.close()
    # lag: the newest queued outputs, never forcing
    dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
    dm.set_output_lag(True)
    kept = {}
    for b in range(nb):
        dm.submit(blocks[b])
        with torch.cuda.stream(side):
            dm.wait_outputs(side)                    # (a no-op behind the first block: nothing is queued yet)
            assert dm.outputs_block() == b - 1
            if b >= 1:
                kept[b - 1] = dm.audio_tensor().clone()
                dm.release_outputs(side)
    dm.synchronize(); side.synchronize()
    for b, t in kept.items():
        assert same(t, b), b
    assert len(kept) == nb - 1
    assert np.array_equal(dm.audio().view(np.uint32), want[-1].view(np.uint32))
    dm.close()
    # fmd_submit_* and fmd_process_* mixed, a control change and a state snapshot in between: the put-off stages are queued first
    dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
    for b in range(nb):
        if b in (2, 5, 6):
            dm.process(blocks[b])
            assert np.array_equal(dm.audio().view(np.uint32), want[b].view(np.uint32)), b
        else:
            dm.submit(blocks[b])
        if b == 3:
            blob = dm.get_state(17)                  # (synchronises: block 3 completes)
            assert np.array_equal(dm.audio().view(np.uint32), want[3].view(np.uint32))
            dm.set_state(17, blob)
        if b == 4:
            dm.set_controls(pkg.default_controls())  # same values: applied at the next block boundary, behind a full drain
    dm.synchronize()
    assert np.array_equal(dm.audio().view(np.uint32), want[-1].view(np.uint32))
    dm.close()


@pytest.mark.parametrize("fs,fast", [(256_000, False), (1_024_000, False), (256_000, True), (1_024_000, True)])
def test_state_snapshot_moves_a_station_between_handles(pkg, fs, fast):
    """fmd_get_state / fmd_set_state: a station's state taken from channel 1 of one handle after 3 blocks and restored into
    channel 4 of another handle (different batch size, different block parity) continues bit-identically."""
    bs = fs * 32 // 1000
    nb = 7
    caps = _caps(3, nb * bs, float(fs), seed=7300)
    a = pkg.BatchDemod(3, bs, fs, keep_taps=True, fast_math=fast)
    for b in range(3):
        a.process(caps[:, b * bs:(b + 1) * bs])
    blob = a.get_state(1)
    assert len(blob) == pkg.load_library().fmd_state_size(a.h)
    other = _caps(6, 2 * bs, float(fs), seed=7400)
    bdm = pkg.BatchDemod(6, bs, fs, keep_taps=True, fast_math=fast)
    for b in range(2):                                   # two blocks: the opposite history-buffer parity
        bdm.process(other[:, b * bs:(b + 1) * bs])
    bdm.set_state(4, blob)
    for b in range(3, nb):
        a.process(caps[:, b * bs:(b + 1) * bs])
        x = np.ascontiguousarray(other[:, :bs]).copy()
        x[4] = caps[1, b * bs:(b + 1) * bs]
        bdm.process(x)
        for name in ("audio", "pll_dt", "rds", "lmr_phase"):
            ga = a.audio() if name == "audio" else a.stream(name)
            gb = bdm.audio() if name == "audio" else bdm.stream(name)
            assert np.array_equal(ga[1].view(np.uint32), gb[4].view(np.uint32)), (name, b)
        sa, ca = a.rds_symbols(); sb, cb = bdm.rds_symbols()
        assert ca[1] == cb[4] and np.array_equal(sa[1, :ca[1]].view(np.uint32), sb[4, :cb[4]].view(np.uint32))
        ba, na = a.rds_bytes(); bb, nb_ = bdm.rds_bytes()
        assert na[1] == nb_[4] and np.array_equal(ba[1, :na[1]], bb[4, :nb_[4]])
    # a blob of another rate is refused
    c = pkg.BatchDemod(1, 2048 if fs != 256_000 else 8192, 256_000 if fs != 256_000 else 1_024_000)
    with pytest.raises(pkg.FmdError):
        c.set_state(0, blob)
    # ... and so is a blob of the other arithmetic mode (the tolerance mode keeps a longer input history)
    e = pkg.BatchDemod(1, bs, fs, fast_math=not fast)
    with pytest.raises(pkg.FmdError):
        e.set_state(0, blob)
    a.close(); bdm.close(); c.close(); e.close()


def test_dead_and_pilotless_channels_do_not_change_their_neighbours(pkg):
    """A band scan's normal case: mono stations, empty channels and all-zero inputs (pilot AGC 1/0, reference agc.h:12-19)
    mixed into a batch of locked stations.  Every healthy channel produces exactly what it produces alone; the dead channels
    produce what the oracle produces for them (NaNs included)."""
    n, bs = 10 * 16384, 16384
    good = _caps(3, n, 256_000.0, seed=8100)
    nopilot = synth.to_cf32(synth.fm_capture(n, fs=256_000.0, seed=8200, pilot_level=0.0)["iq"])
    rng = np.random.default_rng(3)
    noise = (2.0 * rng.standard_normal((n, 2))).astype(np.float32)
    zero = np.zeros((n, 2), np.float32)
    # (a dead front end whose zeros carry signs — 0 x a signed factor, what bench.py's dead channels are: atan2(+-0, +-0) is 0 or +-pi, the
    #  discriminator turns that into full-scale impulses, the pilot loop never locks: no NaN anywhere, every bit the oracle's)
    szero = np.where(rng.random((n, 2)) < 0.5, np.float32(0.0), np.float32(-0.0)).astype(np.float32)
    caps = np.stack([good[0], nopilot, good[1], zero, noise, good[2], zero, szero])
    rep = compare_with_oracle(pkg, caps, bs, 256_000)
    bad = {k: v for k, v in rep["bit_exact"].items() if not v}
    # NaN payloads are not part of the contract: a stream of a dead channel counts as equal when both sides are NaN in the same places
    assert all(k[0] in (3, 6) for k in bad), bad
    alone = run_gpu(pkg, good, bs, 256_000)
    mixed = run_gpu(pkg, caps, bs, 256_000)
    for i, c in enumerate((0, 2, 5)):
        assert np.array_equal(alone["audio"][i].view(np.uint32), mixed["audio"][c].view(np.uint32))
        assert np.array_equal(alone["rds_bytes"][i], mixed["rds_bytes"][c])


def test_bench_two_ranks_as_the_driver_invokes_it(pkg, tmp_path):
    """`python bench.py --gpus 2` with no launcher (the way the driver's SCALE run calls it): bench.py starts its own ranks as a
    child process, RCCL when two devices are visible, otherwise both ranks on cuda:0 over gloo (plumbing only)."""
    import torch
    two = torch.cuda.device_count() >= 2
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--preroll", "2", "--channels", "192",
           "--no-cpu-baseline"] + ([] if two else ["--share-gpu", "--backend", "gloo"])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak" and out["value"] > 0
    assert out["gather_verified"] is True
