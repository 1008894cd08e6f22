"""GPU tests (pytest -m gpu): the tolerance mode against the CPU oracle on REALISTIC captures.

Every other GPU parity capture is oracle/synth.py:fm_capture — two sine tones per channel below 5 kHz, a clean pilot, sigma = 0.02
noise.  The tolerance mode's approximations (pilot phase sampled at 8 points per span behind a 17-tap boxcar, bf16 x 3 operand splits,
the harmonic mixers behind the decimating FIRs with the NCO's deviation taken at the window's centre, in-tile de-emphasis) are statements
about signal statistics, and the reference's own recordings are a release asset that is not in the tree (reference README.md:56-60).
So the synthesiser supplies the realism (oracle/synth.py:fm_capture_realistic): a noise-like programme to 15 kHz with 50 / 75 us
pre-emphasis and independent L / R, peak deviation 75 kHz and an over-deviated case, carrier offsets of +-30 kHz, CNR 15 / 25 / 40 dB,
Rician fading, a pilot 2 Hz off, an adjacent station 200 kHz away (1.024 MSa/s), RDS traffic of three group types.

Per condition, 32 stations x 10 s, u8 captures (the reference's format, src/app.cpp:56-62): against the oracle
  * L+R: whole-run RMS <= 1e-4 on every station (it does not depend on the pilot loop);
  * audio and L-R: whole-run RMS <= 1e-4 over the blocks in which the ORACLE's pilot loop holds lock (RMS of its phase detector
    below 0.25 rad over the block, from 1 s on) — where it does not, the reference's stereo image is noise in any evaluation;
  * RDS: the bit streams identical from lock on wherever the oracle's own stream decodes to the synthesised groups, and the tolerance
    mode decodes as many of them (+-2 %); at CNR 15 dB and through fades, where single symbols are decided by the noise in any
    evaluation: the same groups (>= 95 %) and >= 97 % of the bits, chunk by chunk (_check).
The loop under test: reference src/fm_demod/broadcast_fm_demod.cpp:426-456 (pilot PLL), :463-536 (mixers and decimators),
src/fm_demod/bpsk_synchroniser.cpp:94-186.  Figures go to profiles/round5/parity_metrics.json (and gpurun_out/, which is what
travels back from the GPU box).  The oracle runs are spread over the host's cores."""
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

N_ST, SECONDS = 32, 10.0
LOCK_RAD, LOCK_FROM_S = 0.25, 1.0


def _station(args):
    """(worker process) capture of one station and its oracle outputs with the library's coefficients"""
    cond, kw, n_blocks, seed, c, coeff_bytes, fs, bs = args
    import oraclelib as O
    k = O.Coeffs.from_buffer_copy(coeff_bytes)
    cap_c = synth.fm_capture_realistic(n_blocks * bs, fs=float(fs), seed=seed, channel=c, **kw)
    cap = synth.to_u8(cap_c["iq"])
    o = O.run_chain(cap, bs, fs, u8=True, coeffs=k, streams=["lpr", "lmr", "audio", "lmr_phase", "pll_raw_err"])
    n_fo = o["pll_raw_err"].size // n_blocks
    lock = np.sqrt((o["pll_raw_err"].astype(np.float64).reshape(n_blocks, n_fo) ** 2).mean(axis=1))
    return c, cap, o["lpr"], o["lmr"], o["audio"], o["lmr_phase"], o["rds_bytes"], lock, cap_c["groups"]


@pytest.fixture(scope="module")
def pkg():
    import fmradio_loader
    p = fmradio_loader.load()
    p.load_library()
    import torch
    assert torch.cuda.is_available()
    return p


def _run_condition(pkg, cond, fs, bs):
    import test_gpu_fast as F
    from rds_groups import decode_groups
    kw = synth.REALISTIC_CONDITIONS[cond]
    n_blocks = int(SECONDS * fs / bs)
    dm = pkg.BatchDemod(N_ST, bs, fs, keep_taps=True, fast_math=True)
    coeff = bytes(dm.get_coeffs(0))
    workers = min(N_ST, max(1, (os.cpu_count() or 8) // 2))
    seed = 7000 + 37 * sorted(synth.REALISTIC_CONDITIONS).index(cond)
    with ProcessPoolExecutor(workers) as ex:
        res = sorted(ex.map(_station, [(cond, kw, n_blocks, seed, c, coeff, fs, bs) for c in range(N_ST)]), key=lambda r: r[0])
    n_a = bs // (fs // 256_000) // 8
    sq = {k: np.zeros((N_ST, n_blocks)) for k in ("lpr", "lmr", "audio")}
    pw = {k: np.zeros((N_ST, n_blocks)) for k in ("lpr", "lmr", "audio")}
    rds = [b"" for _ in range(N_ST)]
    for b in range(n_blocks):
        blk = np.stack([r[1][b * bs:(b + 1) * bs] for r in res])
        assert dm.process(np.ascontiguousarray(blk)) == 0
        g = {"audio": dm.audio().astype(np.float64).reshape(N_ST, -1), "lpr": dm.stream("lpr").astype(np.float64), "lmr": dm.stream("lmr").astype(np.float64)}
        by, bc = dm.rds_bytes()
        for c in range(N_ST):
            for k, idx, w in (("lpr", 2, n_a), ("lmr", 3, n_a), ("audio", 4, 2 * n_a)):
                ref = res[c][idx][b * w:(b + 1) * w].astype(np.float64)
                sq[k][c, b] = np.mean((g[k][c] - ref) ** 2)
                pw[k][c, b] = np.mean(ref ** 2)
            rds[c] += by[c, :bc[c]].tobytes()
    dm.close()
    lock = np.stack([r[7] for r in res])                                   # [C, n_blocks] RMS of the oracle's phase detector (rad)
    first = int(np.ceil(LOCK_FROM_S * fs / bs))
    locked = (lock < LOCK_RAD) & (np.arange(n_blocks)[None, :] >= first)
    out = {"condition": cond, "fs": fs, "stations": N_ST, "seconds": SECONDS, "blocks": n_blocks, "capture": {k: v for k, v in kw.items()},
           "locked_station_blocks": int(locked.sum()), "station_blocks": int(N_ST * n_blocks),
           "stations_locked_over_90pct": int(np.sum(locked[:, first:].mean(axis=1) > 0.9))}
    out["lpr_rms_worst"] = float(np.sqrt(sq["lpr"].mean(axis=1)).max())
    out["lpr_signal_rms"] = float(np.sqrt(pw["lpr"].mean()))
    for k in ("lmr", "audio"):
        per = [np.sqrt(sq[k][c][locked[c]].mean()) for c in range(N_ST) if locked[c].sum() >= 16]
        out[k + "_rms_worst_in_lock"] = float(max(per)) if per else None
        out[k + "_rms_median_in_lock"] = float(np.median(per)) if per else None
        out[k + "_worst_block_in_lock"] = float(np.sqrt(sq[k][locked].max())) if locked.any() else None
        out[k + "_signal_rms"] = float(np.sqrt(pw[k].mean()))
        out[k + "_rms_worst_whole_run"] = float(np.sqrt(sq[k].mean(axis=1)).max())
    # RDS: where the oracle's own stream decodes, the two streams carry the same bits from lock on
    n_dec = n_same = 0
    ratio, agree = [], []
    for c in range(N_ST):
        want = {tuple(int(v) for v in w) for w in res[c][8]}
        go = [w for w in decode_groups(res[c][6]) if w in want]
        gg = [w for w in decode_groups(np.frombuffer(rds[c], np.uint8)) if w in want]
        if len(go) >= 0.8 * (SECONDS - 1.0) * 1187.5 / 104:               # the oracle decodes this station (>= 80 % of the groups sent behind the first second)
            n_dec += 1
            n_same += int(F.same_bits_once_in_lock(np.frombuffer(rds[c], np.uint8), res[c][6], skip_bits=8 * 76))
            ratio.append(len(gg) / max(len(go), 1))
            agree.append(bit_agreement(np.frombuffer(rds[c], np.uint8), res[c][6], skip_bits=8 * 76))
    out["rds_stations_the_oracle_decodes"] = n_dec
    out["rds_stations_with_identical_bits_from_lock"] = n_same
    out["rds_groups_decoded_vs_oracle_min"] = float(min(ratio)) if ratio else None
    out["rds_bit_agreement_min"] = float(min(agree)) if agree else None
    out["rds_bit_agreement_mean"] = float(np.mean(agree)) if agree else None
    return out, F


def bit_agreement(a, b, skip_bits, chunk=256, reach=32):
    """Fraction of the bits of stream a (behind skip_bits) that stream b carries too, chunk by chunk, each chunk at the best shift within
    `reach` of the previous chunk's: a slip of the synchroniser in one run (a fade, a noise burst) moves the alignment once, it does not
    make every later bit "different".  1.0 = identical from lock on (what same_bits_once_in_lock asks for, without its single alignment)."""
    a = np.unpackbits(np.asarray(a, np.uint8)); b = np.unpackbits(np.asarray(b, np.uint8))
    pos, shift, same, total = skip_bits, 0, 0, 0
    while pos + chunk <= a.size:
        best, best_sh = -1, shift
        for sh in range(shift - reach, shift + reach + 1):
            lo = pos + sh
            if lo < 0 or lo + chunk > b.size:
                continue
            eq = int(np.sum(a[pos:pos + chunk] == b[lo:lo + chunk]))
            if eq > best or (eq == best and abs(sh - shift) < abs(best_sh - shift)):
                best, best_sh = eq, sh
        if best < 0:
            break
        same += best; total += chunk; shift = best_sh
        pos += chunk
    return same / total if total else 0.0


def _record(F, name, out):
    """straight into profiles/round5 (the tracked copy) and into gpurun_out/ (what travels back from the GPU box)"""
    import json, pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    for path in (root / "profiles" / "round5" / "parity_metrics.json", root / "gpurun_out" / "parity_metrics_round5.json"):
        try:
            path.parent.mkdir(parents=True, exist_ok=True)
            doc = json.loads(path.read_text()) if path.exists() else {}
            doc[name] = out
            path.write_text(json.dumps(doc, indent=1, sort_keys=True))
        except Exception as e:      # noqa: BLE001
            print("parity metrics not recorded:", e)


def _check(out, F, rds_required=True, rds_identical=True):
    """rds_identical: the bit streams identical from lock on, on every station the oracle decodes.  Not asked of the two conditions in which the
    RDS subcarrier itself goes through the noise (CNR 15 dB; fades): there the synchroniser's decisions on single symbols are the noise's, the
    two runs err on different symbols and slip at different fades — as two builds of the reference do (profiles/round4/reference_flip_evidence.json) —
    and the bar is what a listener's decoder sees: the same groups decoded (>= 95 % of the oracle's count) and >= 97 % of the bits agreeing
    chunk by chunk on the worst station (bit_agreement; a slip costs the chunk it falls in)."""
    print(out)
    assert out["lpr_rms_worst"] <= F.TOL_RMS, out
    if out["lmr_rms_worst_in_lock"] is not None:
        assert out["lmr_rms_worst_in_lock"] <= F.TOL_RMS, out
        assert out["audio_rms_worst_in_lock"] <= 2 * F.TOL_RMS, out           # audio = 2 (L+R +- L-R): twice the rails' bar, as in tests/test_gpu_fast.py
    if rds_required:
        assert out["rds_stations_the_oracle_decodes"] >= N_ST // 2, out           # the comparison is not vacuous
    if rds_identical:
        assert out["rds_stations_with_identical_bits_from_lock"] == out["rds_stations_the_oracle_decodes"], out
        if out["rds_groups_decoded_vs_oracle_min"] is not None:
            assert out["rds_groups_decoded_vs_oracle_min"] >= 0.98, out
    elif out["rds_groups_decoded_vs_oracle_min"] is not None:
        assert out["rds_groups_decoded_vs_oracle_min"] >= 0.95, out
        assert out["rds_bit_agreement_min"] >= 0.97, out            # (a slip costs the chunk it falls in: ~1 % of a 10 s run)
        assert out["rds_stations_with_identical_bits_from_lock"] >= out["rds_stations_the_oracle_decodes"] // 2, out


@pytest.mark.parametrize("cond", ["programme_cnr40", "programme_cnr25", "programme_cnr15", "preemphasis_50us", "carrier_plus_30k", "carrier_minus_30k",
                                  "overdeviation_110k", "fading_5hz", "pilot_plus_2hz"])
def test_tolerance_mode_on_realistic_captures_256k(pkg, cond):
    out, F = _run_condition(pkg, cond, 256_000, 16384)
    _record(F, f"realistic_256k_u8_{cond}", out)
    _check(out, F, rds_identical=cond not in ("programme_cnr15", "fading_5hz"))


@pytest.mark.parametrize("cond", ["programme_cnr40", "programme_cnr25", "carrier_plus_30k", "adjacent_minus_20db", "fading_5hz"])
def test_tolerance_mode_on_realistic_captures_at_the_reference_rate(pkg, cond):
    """1.024 MSa/s u8, the reference's own rate and format (broadcast_fm_demod.cpp:62-77): first decimator included; the adjacent station
    200 kHz away only fits this bandwidth."""
    out, F = _run_condition(pkg, cond, 1_024_000, 65536)
    _record(F, f"realistic_1024k_u8_{cond}", out)
    _check(out, F, rds_identical=cond not in ("fading_5hz",))
