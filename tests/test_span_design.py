"""CPU test (no GPU): the tables of the tolerance mode's span-wise pilot PLL (k_pll_span) as the library designs them
(fmd_design_pll_span, host code) against an independent float64 restatement of the same linear map, and that map against the
reference's loop run sample by sample (reference broadcast_fm_demod.cpp:430-456) on random error sequences."""
import ctypes as C

import numpy as np

import oraclelib as O

L = 128
N1, N2 = 41, 84
TWO_PI = 2.0 * np.pi


def _lib():
    import fmradio_loader
    return fmradio_loader.load().load_library()


def _tables():
    w = np.zeros((5, L), np.float32); s = np.zeros((5, 8), np.float32); minv = np.zeros((3, 4), np.float32); misc = np.zeros(2, np.float32)
    rc = _lib().fmd_design_pll_span(256000, w.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), minv.ctypes.data_as(C.c_void_p), misc.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return w, s, minv, misc


def _serial(k, lpf, I, e1, e2, r0, eh_turns, Ts):
    """The reference loop over one span with the NCO HELD at F0 = f[0] + r0 as the hold the tables assume: returns the state after
    the span and the phase deviation from the hold at every sample."""
    b0, b1, a0 = float(k.pll_lpf_b[0]), float(k.pll_lpf_b[1]), float(k.pll_lpf_a[0])
    ktsi = float(np.float32(0.1) * np.float32(Ts))
    dev = np.zeros(L)
    d = 0.0
    for n in range(L):
        lpf = b0 * e2 + b1 * e1 + a0 * lpf
        I = I + ktsi * e1
        f = -19000.0 - 100.0 * (0.01 * lpf + I)
        if n == 0:
            F0 = f + r0
        d += Ts * (f - F0)
        dev[n] = d
        e2, e1 = e1, TWO_PI * (eh_turns[n] + d)
    return lpf, I, dev


def test_span_tables_reproduce_the_serial_loop():
    w, s, minv, misc = _tables()
    k = O.design(256000)
    Ts = float(np.float32(1.0) / np.float32(128000.0))
    rng = np.random.default_rng(3)
    for trial in range(20):
        lpf, I, e1, e2 = rng.normal(0, 0.05), rng.normal(0, 1e-3), rng.normal(0, 0.1), rng.normal(0, 0.1)
        r0 = rng.uniform(-1e-3, 1e-3)
        eh = 0.02 * np.cumsum(rng.normal(0, 0.02, L)) + 0.02 * np.sin(np.arange(L) * 0.05) + rng.uniform(-1e-3, 1e-3, L)   # a slowly moving error + detector noise
        want_lpf, want_I, dev = _serial(k, lpf, I, e1, e2, r0, eh, Ts)
        v = np.array([lpf, I, e1, e2, r0])
        rows = w.astype(np.float64) @ eh + s[:, :5].astype(np.float64) @ v
        assert abs(rows[0] - want_lpf) <= 2e-6 * max(1.0, abs(want_lpf)), (trial, rows[0], want_lpf)
        assert abs(rows[1] - want_I) <= 1e-9 + 2e-6 * abs(want_I), (trial, rows[1], want_I)
        for r, n in ((2, N1), (3, N2), (4, L - 1)):
            assert abs(rows[r] - dev[n]) <= 1e-9 + 2e-6 * abs(dev[n]), (trial, r, rows[r], dev[n])
        # the cubic through the three deviation rows reproduces the deviation at every sample of the span
        abg = minv[:, :3].astype(np.float64) @ rows[2:5]
        nn = np.arange(L, dtype=np.float64)
        fit = abg[0] * nn + abg[1] * nn ** 2 + abg[2] * nn ** 3
        # (random start states put an exponential of the loop filter, 200 samples long, into the span: the worst case for a cubic)
        assert np.max(np.abs(fit - dev)) <= 1.5e-6 + 0.05 * np.max(np.abs(dev)), (trial, np.max(np.abs(fit - dev)), np.max(np.abs(dev)))


def test_quadrature_factor_is_the_hilbert_fir_at_the_pilot():
    _, _, _, misc = _tables()
    k = O.design(256000)
    h = np.array(k.b_hilbert, np.float64)
    w0 = TWO_PI * 19000.0 / 128000.0
    H = np.sum(h * np.exp(-1j * w0 * np.arange(65)))
    # for s[n] = cos(w0 n) the reference's Hilbert rail im[n] = sum_k b[k] s[n - 64 + k] (hilbert_fir_filter.h:26-46) beside re[n] = s[n - 32]
    # must equal quad (re[n-1] - re[n+1])
    n = np.arange(200, 400)
    y = np.array([np.sum(h * np.cos(w0 * (i - 64 + np.arange(65)))) for i in n])
    approx = float(misc[0]) * (np.cos(w0 * (n - 33)) - np.cos(w0 * (n - 31)))
    assert np.max(np.abs(y - approx)) <= 1e-5 * abs(H)
    Ts = float(np.float32(1.0) / np.float32(128000.0))
    assert abs(float(misc[1]) - (-19000.0 * Ts + 19.0 / 128.0)) < 1e-12
