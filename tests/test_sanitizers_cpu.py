"""CPU sanitizer run (SURVEY §4 "sanitizers on the CPU build"; never on the GPU box's device code): the oracle (oracle/fm_oracle.c),
the library's host-side filter designer (fm-radio_amd/csrc/fmd_design.cpp) and the host-side drivers (group synchroniser, scraper
writers) built with -fsanitize=address,undefined (`make -C oracle asan`) and fed the golden fixtures.  Any out-of-bounds access, use after
free, leak, signed overflow or misaligned access aborts the run (-fno-sanitize-recover); the outputs must still be the fixtures'.

The reference's own trouble spots on this path: a static lambda capture in its designer (src/dsp/filter_designer.cpp:235,288,345) and
the re-blocking buffer's span arithmetic (src/utility/reconstruction_buffer.h:16-26) — the driver re-blocks ragged pieces the same way."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oraclelib as O

ROOT = Path(__file__).resolve().parent.parent
ASAN = ROOT / "oracle" / "_asan"
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


@pytest.fixture(scope="module")
def asan_build():
    r = subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return ASAN


def _run(cmd):
    r = subprocess.run([str(c) for c in cmd], env=ENV, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    return r


@pytest.mark.parametrize("fixture,u8,bs,fs", [("chain_b16384", True, 16384, 1_024_000), ("chain_cf32_b8192", False, 8192, 1_024_000)])
def test_oracle_under_sanitizers_reproduces_the_golden_chain(asan_build, tmp_path, fixture, u8, bs, fs):
    g = np.load(ROOT / "tests" / "golden" / f"{fixture}.npz")
    cap = tmp_path / "cap.bin"
    np.ascontiguousarray(g["capture"]).tofile(cap)
    _run([asan_build / "sanitizer_main", "chain", cap, "u8" if u8 else "cf32", bs, fs, tmp_path / "out"])
    audio = np.fromfile(str(tmp_path / "out") + ".audio.f32", np.float32)
    counts = np.fromfile(str(tmp_path / "out") + ".rds_count.i32", np.int32)
    phase = np.fromfile(str(tmp_path / "out") + ".lmr_phase.f32", np.float32)
    # the fixtures were dumped from the compiled reference (tests/golden/make_golden.py); its pilot gain goes through rsqrtss, which
    # differs between CPUs: the north-star tolerance here, bit-identity is tests/test_oracle_vs_ref.py's business
    nb = g["capture"].shape[0] // bs
    assert audio.size == g["audio"].size
    assert np.sqrt(np.mean((audio.astype(np.float64) - g["audio"]) ** 2)) <= 1e-4
    assert counts.size == nb and np.array_equal(counts, g["rds_count"])
    assert np.allclose(phase[-g["lmr_phase"].size:], g["lmr_phase"], atol=1e-4)


def test_designer_under_sanitizers_equals_the_oracle_design(asan_build, tmp_path):
    for fs in (256_000, 1_024_000, 2_048_000):
        out = tmp_path / f"coeffs_{fs}.bin"
        _run([asan_build / "sanitizer_main", "design", fs, out])
        k = O.Coeffs.from_buffer_copy(out.read_bytes())
        want = O.design(fs, rsqrt_mode=0)
        for name in ("b_fm_in", "b_fm_out", "b_hilbert", "pll_lpf_b", "pll_lpf_a", "b_lpr", "b_lmr", "b_rds", "ted_lpf_b", "bpsk_lpf_b"):
            a, b = k.arr(name), want.arr(name)
            assert np.array_equal(a, b), (fs, name, np.abs(a - b).max())
        assert np.allclose(k.arr("pilot_b"), want.arr("pilot_b"), rtol=3e-7) and np.array_equal(k.arr("pilot_a"), want.arr("pilot_a"))
        k2 = O.Coeffs.from_buffer_copy(Path(str(out) + ".ctl").read_bytes())
        ctl = O.default_controls(); ctl.use_deemphasis, ctl.deemphasis_tus, ctl.lpr_cutoff_hz = 1, 50, 12000
        want2 = O.design(fs, controls=ctl, rsqrt_mode=0)
        for name in ("deemph_b", "deemph_a", "b_lpr", "b_lmr"):
            assert np.array_equal(k2.arr(name), want2.arr(name)), (fs, name)


def test_host_drivers_under_sanitizers(asan_build, tmp_path):
    g = np.load(ROOT / "tests" / "golden" / "long_b65536.npz")
    # the RDS group synchroniser on the golden byte stream: the groups the reference's decoder logged (tests/golden/make_golden.py)
    (tmp_path / "rds.bin").write_bytes(g["rds_bytes"].tobytes())
    out = _run([asan_build / "group_sync_main", tmp_path / "rds.bin"]).stdout.splitlines()
    groups = [l for l in out if l.startswith("[group]")]
    want = ["[group] [" + " ".join(f"{int(v):04X}" for v in row) + "]" for row in g["groups"]]
    assert len(groups) >= 15 and all(x in want for x in groups if "----" not in x)
    # the scraper-compatible writers on golden audio blocks
    audio = g["audio"].astype(np.float32)                                    # [3][4096] = 3 blocks of 2048 frames
    audio.tofile(tmp_path / "audio.f32")
    _run([asan_build / "scraper_writer_main", tmp_path / "audio.f32", tmp_path / "rds.bin", 2048, tmp_path / "o.wav", tmp_path / "o.bin"])
    wav = (tmp_path / "o.wav").read_bytes()
    assert wav[:4] == b"RIFF" and wav[8:12] == b"WAVE" and len(wav) == 44 + audio.size * 2
    pcm = np.frombuffer(wav[44:], np.int16)
    assert np.array_equal(pcm, (audio.reshape(-1) * (np.float32(32767.0) * np.float32(0.95))).astype(np.int32).astype(np.int16))
    assert (tmp_path / "o.bin").read_bytes() == g["rds_bytes"].tobytes()[:len(g["rds_bytes"]) // 16 * 16]
