"""CPU test (no GPU): the composite band-pass FIRs of the tolerance mode's extract stage (k_extract_bp, fmd_kernels_bp.inc) as the library
designs them (fmd_design_extract_bp, host code) against the reference's order of operations restated in float64: mix every 128 kHz sample
of the analytic signal with the NCO's harmonic, then low-pass and decimate (reference broadcast_fm_demod.cpp:463-536,
apply_harmonic_pll.cpp:88-139, hilbert_fir_filter.h:26-46).  With the loop's deviation p constant the two are the same linear map; with a
deviation that moves at f Hz the composite form sees its filter shifted by H f — bounded here at the figures the kernel's header quotes."""
import ctypes as C

import numpy as np
import pytest

import oraclelib as O

NT = 192


def _lib():
    import fmradio_loader
    return fmradio_loader.load().load_library()


def _design(cutoff_hz=15000):
    g2 = np.zeros((2, NT), np.float32); g3 = np.zeros((2, NT), np.float32)
    L = _lib()
    L.fmd_design_extract_bp.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert L.fmd_design_extract_bp(256000, cutoff_hz, g2.ctypes.data_as(C.c_void_p), g3.ctypes.data_as(C.c_void_p)) == 0
    return g2[0].astype(np.float64) + 1j * g2[1], g3[0].astype(np.float64) + 1j * g3[1]


def _reference_order(x, h, hil, H, M, p):
    """y[m] = sum_tau h[tau] a[t0 + tau] e^{j 2 pi H dt[t0 + tau]}, t0 = M (m + 1) - 128, dt[t] = p[t] - 19 (t + 1) / 128; x[s] = 0 for s < 0."""
    n = x.size
    xp = np.concatenate([np.zeros(64), x])
    im = np.array([np.dot(xp[i:i + 65], hil) for i in range(n)])
    a = np.concatenate([np.zeros(32), x])[:n] + 1j * im
    t = np.arange(n, dtype=np.float64)
    mix = a * np.exp(2j * np.pi * H * (p - 19.0 * (t + 1.0) / 128.0))
    out = []
    for m in range(n // M):
        t0 = M * (m + 1) - 128
        if t0 < 0:
            continue
        out.append((m, np.dot(mix[t0:t0 + 128], h)))
    return out


def _composite(x, G, H, M, p, m):
    """the kernel's form for output m: e^{j 2 pi (H p(t_c) - H 19 (t0 + 1) / 128)} sum_u G[u] x[t0 - 64 + u]"""
    t0 = M * (m + 1) - 128
    s = np.dot(G, np.concatenate([np.zeros(64), x])[t0:t0 + NT])       # x[t0 - 64 + u]
    pc = 0.5 * (p[t0 + 63] + p[t0 + 64])
    return s * np.exp(2j * np.pi * (H * pc - H * 19.0 * (t0 + 1.0) / 128.0))


@pytest.mark.parametrize("cutoff", [15000, 9000])
def test_composite_taps_are_the_mixer_and_hilbert_fir_folded_into_the_decimator(cutoff):
    ctl = O.default_controls()
    ctl.lmr_cutoff_hz = cutoff
    k = O.design(256000, ctl)
    G2, G3 = _design(cutoff)
    rng = np.random.default_rng(7)
    x = rng.standard_normal(4096) * 0.2
    hil = k.arr("b_hilbert").astype(np.float64)
    for G, h, H, M in ((G2, k.arr("b_lmr"), 2, 4), (G3, k.arr("b_rds"), 3, 8)):
        p = np.full(x.size, 0.1234)                                     # a constant deviation: the two forms are the same map
        ref = _reference_order(x, h.astype(np.float64), hil, H, M, p)
        err = max(abs(_composite(x, G, H, M, p, m) - y) for m, y in ref)
        scale = np.sqrt(np.mean([abs(y) ** 2 for _, y in ref]))
        assert err <= 2e-6 * max(scale, 1e-3), (H, err, scale)           # float taps against float64 ones


def test_a_moving_deviation_shifts_the_filter_by_h_times_its_rate():
    """NCO 2 Hz off the nominal pilot (a transmitter at the edge of its +-2 Hz tolerance): L-R within 1e-4 of the signal's scale (4e-6
    absolute per Hz),
    the RDS rails within 3e-3 of theirs (their filter's transition band is where the RDS spectrum lives)."""
    k = O.design(256000)
    G2, G3 = _design()
    rng = np.random.default_rng(8)
    n = 8192
    t = np.arange(n) / 128000.0
    # an MPX-like signal: audio tones, pilot, L-R on 38 kHz, RDS-like BPSK on 57 kHz
    bits = np.repeat(rng.integers(0, 2, n // 54 + 1) * 2.0 - 1.0, 54)[:n]
    x = 0.2 * np.sin(2 * np.pi * 1000 * t) + 0.05 * np.sin(2 * np.pi * 19000 * t) + 0.15 * np.sin(2 * np.pi * 3000 * t) * np.sin(2 * np.pi * 38000 * t) \
        + 0.03 * bits * np.sin(2 * np.pi * 57000 * t) + 0.002 * rng.standard_normal(n)
    hil = k.arr("b_hilbert").astype(np.float64)
    p = 0.05 + 2.0 * t                                                   # 2 Hz
    for G, h, H, M, bar in ((G2, k.arr("b_lmr"), 2, 4, 1e-4), (G3, k.arr("b_rds"), 3, 8, 3e-3)):
        ref = _reference_order(x, h.astype(np.float64), hil, H, M, p)[40:]
        err = np.sqrt(np.mean([abs(_composite(x, G, H, M, p, m) - y) ** 2 for m, y in ref]))
        scale = np.sqrt(np.mean([abs(y) ** 2 for _, y in ref]))
        assert err <= bar * scale, (H, err, scale)
