"""The oracle against the compiled reference itself (oracle/_ref/fm_ref_dump), live.

Runs wherever the prebuilt _ref binaries exist (they are built from /root/reference by
`make -C oracle ref` and travel with the snapshot); skipped otherwise.  Every stream the
reference exposes must be bit-identical to the restatement on the same input.
"""
import numpy as np
import pytest

import oraclelib as O
import synth
from conftest import bits_equal, describe_diff

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference here)")

STREAMS = ["fm_out_iq", "pilot", "pll", "pll_raw_err", "pll_pi_err", "lpr", "lmr", "lmr_phase", "rds", "rds_raw_sym", "rds_sym",
           "audio", "bpsk_pll_sym", "bpsk_intdump", "bpsk_ted_raw", "bpsk_ted_pi", "bpsk_pll_raw", "bpsk_pll_pi"]


def _assert_same(ref, orc):
    for k in STREAMS:
        assert bits_equal(ref[k], orc[k]), f"{k}: {describe_diff(ref[k], orc[k])}"
    assert np.array_equal(ref["rds_count"], orc["rds_count"])
    for k in ("bpsk_zcd", "bpsk_trig"):      # the reference's two bool traces (bpsk_synchroniser.h:79-80); the oracle hands them out as 0 / 1 floats
        assert np.array_equal(ref[k].astype(np.float32), orc[k]), k
    assert np.array_equal(ref["rds_bytes"], orc["rds_bytes"])


@pytest.fixture(scope="module")
def capture_u8():
    return synth.to_u8(synth.fm_capture(24 * 65536, seed=2024)["iq"])


@pytest.mark.parametrize("block_size,n_blocks", [(65536, 24), (16384, 40), (8192, 24), (131072, 8)])
def test_chain_bit_exact(tmp_path, capture_u8, block_size, n_blocks):
    cap = capture_u8[: block_size * n_blocks]
    _assert_same(O.run_ref_chain(cap, tmp_path, block_size), O.run_chain(cap, block_size))


def _ctl(**kw):
    c = O.default_controls()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize("args,ctl", [
    (["deemph=50"], dict(use_deemphasis=1, deemphasis_tus=50)),
    (["deemph=75", "audio=lmr"], dict(use_deemphasis=1, deemphasis_tus=75, audio_out=1)),
    (["audio=lpr", "lpr=12000"], dict(audio_out=0, lpr_cutoff_hz=12000)),
    (["mix=0.65", "lmr=9000", "lpr=100"], dict(audio_stereo_mix_factor=0.65, lmr_cutoff_hz=9000, lpr_cutoff_hz=100)),
    (["lpr=70000", "deemph=100"], dict(lpr_cutoff_hz=70000, use_deemphasis=1, deemphasis_tus=100)),
])
def test_controls_bit_exact(tmp_path, capture_u8, args, ctl):
    cap = capture_u8[: 65536 * 6]
    _assert_same(O.run_ref_chain(cap, tmp_path, 65536, extra_args=args), O.run_chain(cap, 65536, controls=_ctl(**ctl)))


def test_cf32_boundary_bit_exact(tmp_path):
    cap = synth.to_cf32(synth.fm_capture(6 * 65536, seed=77)["iq"])
    _assert_same(O.run_ref_chain(cap, tmp_path, 65536, u8=False), O.run_chain(cap, 65536, u8=False))


def test_noise_only_input_bit_exact(tmp_path):
    """No station: PLLs never lock, AGC rails — exercises the unlocked/acquisition dynamics."""
    rng = np.random.default_rng(3)
    cap = np.clip(np.rint(127 + 30 * rng.standard_normal((4 * 65536, 2))), 0, 255).astype(np.uint8)
    _assert_same(O.run_ref_chain(cap, tmp_path, 65536), O.run_chain(cap, 65536))
