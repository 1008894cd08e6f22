"""CPU test (no GPU): the tables of the tolerance mode's span-wise pilot PLL (k_pll_span) as the library designs them
(fmd_design_pll_span, host code) against an independent float64 restatement of the same linear map, and that map against the
reference's loop run sample by sample (reference broadcast_fm_demod.cpp:430-456) on random error sequences."""
import ctypes as C

import numpy as np
import pytest

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


# ---- round 4: the sparse form (k_pll_sparse, fmd_kernels.h PllSparseTab) -------------------------------------------------

def _sparse_tables():
    taps = np.zeros((2, 32), np.float32); cplx = np.zeros((19, 2), np.float32); rows = np.zeros((2, 8), np.float32)
    sw = np.zeros((5, 132), np.float32); misc = np.zeros(8, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert _lib().fmd_design_pll_sparse(256000, p(taps), p(cplx), p(rows), p(sw), p(misc)) == 0
    return taps, cplx, rows, sw, misc


def test_sparse_rows_are_the_span_rows_applied_to_a_line():
    w, s, minv, _ = _tables()
    taps, cplx, rows, sw, misc = _sparse_tables()
    nbar = float(misc[2])
    assert abs(nbar - np.mean(16 * np.arange(8) + 9)) < 1e-6
    n = np.arange(L, dtype=np.float64)
    w64 = w.astype(np.float64)
    w64[2:5] = minv[:, :3].astype(np.float64) @ w64[2:5]              # the sparse form's rows 2..4 are the cubic's coefficients themselves
    assert np.allclose(rows[0, :5], w64.sum(axis=1), rtol=2e-5, atol=1e-9)
    assert np.allclose(rows[1, :5], (w64 * (n - nbar)).sum(axis=1), rtol=2e-5, atol=1e-8)
    suf = np.cumsum(w64[:, ::-1], axis=1)[:, ::-1]
    assert np.allclose(sw[:, :L], suf, rtol=2e-5, atol=1e-9) and np.all(sw[:, L] == 0.0)
    # a line through the 8 points: mean and slope as the kernel takes them reproduce sum_n w[r][n] (a + b (n - nbar))
    nk = 16 * np.arange(8) + 9
    a, b = 0.0123, -3.1e-4
    ek = a + b * (nk - nbar)
    am, bs = ek.mean(), np.sum((nk - nbar) * ek) * float(misc[1])
    want = w64 @ (a + b * (n - nbar))
    got = rows[0, :5].astype(np.float64) * am + rows[1, :5].astype(np.float64) * bs
    assert np.allclose(got, want, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("offset_hz", [0.0, 0.7, -35.0])
def test_sparse_points_see_the_phase_the_reference_filter_has(offset_hz):
    """A pilot tone through the reference's peak filter sample by sample (iir_filter.h:40-69 with the designed float coefficients)
    against the decimated one-pole form of the tables: at every point the phase must be the filter output's (quadrature taken from
    its own neighbours at the tone's frequency, i.e. without the reference's Hilbert-rail gain ripple, which averages out under the
    loop's rows)."""
    taps, cplx, rows, sw, misc = _sparse_tables()
    k = O.design(256000)
    pk, pa0, pa1 = float(k.pilot_b[0]), float(k.pilot_a[0]), float(k.pilot_a[1])
    n = 60000                                                       # six time constants of the filter
    wt = TWO_PI * (19000.0 + offset_hz) / 128000.0
    x = 0.05 * np.cos(wt * np.arange(n) + 0.4)
    xd = np.concatenate([np.zeros(33), x])                          # x'[m - 2] = fm_out[m - 33]
    P = np.zeros(n + 2)
    for i in range(n):
        P[i + 2] = pk * xd[i] + pa1 * P[i + 1] + pa0 * P[i]         # P[i + 2] = P'[i]
    Wc = taps[0].astype(np.float64) + 1j * taps[1].astype(np.float64)
    rot = cplx[:8, 0].astype(np.float64) + 1j * cplx[:8, 1].astype(np.float64)
    rho16 = complex(cplx[8, 0], cplx[8, 1])
    assert abs(complex(cplx[9, 0], cplx[9, 1]) - rho16 ** 2) < 1e-6 and abs(complex(cplx[10, 0], cplx[10, 1]) - rho16 ** 4) < 1e-6
    assert all(abs(complex(cplx[11 + i, 0], cplx[11 + i, 1]) - rho16 ** (i + 1)) < 2e-6 for i in range(8))
    kap2 = complex(misc[5], misc[6])
    xp = np.concatenate([np.zeros(192), x])
    Z = 0j
    worst = 0.0
    for q in range(n // L):
        for kk in range(8):
            base = 192 + L * q + 16 * kk - 48                        # the point's 32 inputs: columns k - 3 and k - 2 of the front end's tiles
            V = rot[kk] * np.sum(Wc * xp[base:base + 32])
            Z = rho16 * Z + V
            if q >= 400:                                             # settled
                nn = L * q + 16 * kk + 9                             # sample index of the point
                phi = np.angle(Z + kap2 * V) / TWO_PI + float(misc[0])          # + phi0
                re, im = P[nn + 1], (P[nn] - P[nn + 2]) / (2.0 * np.sin(wt))      # P'[nn - 1], (P'[nn - 2] - P'[nn]) / (2 sin w)
                psi = np.arctan2(im, re) / TWO_PI - 19.0 * ((16 * kk + 9) + 1) / 128.0
                dlt = phi - psi
                worst = max(worst, abs(dlt - np.round(dlt)))
    assert worst < (2e-6 if abs(offset_hz) < 5 else 6e-6), worst
