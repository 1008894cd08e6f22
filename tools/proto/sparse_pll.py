"""Offline model (float64, numpy) of the round-4 tolerance-mode pilot PLL, "sparse" form, against the CPU oracle
(development tool; tests/-style use of oracle/ — never imported by the product).

Round 3's k_pll_span takes one arctangent per 128 kHz sample of the filtered pilot.  Behind the peak filter (r = 0.9999) the pilot is a
line a few Hz wide: its phase relative to a 19 kHz reference is smooth over a span of 128 samples, so the loop's five weight rows
need it at a handful of points only.  Exact identity used here (no narrow-band assumption):

    P[m] = K x[m-2] + a1 P[m-1] + a0 P[m-2]   (real rail of the reference's peak filter, poles r exp(+-j w0))
         = (K / sin w0) Im{ exp(j w0 (m+1)) Z[m] },     Z[m] = r Z[m-1] + exp(-j w0 m) x[m-2]

i.e. the resonator IS a complex one-pole low-pass of the down-mixed input.  Z decimates exactly:
    Z[m] = r^D Z[m-D] + sum_{i<D} r^(D-1-i) u[m-D+1+i],  u[m] = exp(-j w0 m) x[m-2]
and with w0 = 2 pi 19/128 the mixer is periodic in the span.  What the per-sample arctangent of the reference adds on top of arg Z is
(a) the ellipse of its Hilbert rail (gain |H(w0)| != 1) and (b) the leakage of programme content >= 4 kHz away from the pilot, both of
which the smooth weight rows average out; a linear-phase band-pass in front of the decimation keeps (b) from aliasing.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oraclelib as O  # noqa: E402
import synth  # noqa: E402

L = 128
N1, N2 = 41, 84
TWO_PI = 2 * np.pi
W0 = TWO_PI * 19.0 / 128.0
TS32 = float(np.float32(1.0) / np.float32(128000.0))
KTSI = float(np.float32(0.1) * np.float32(TS32))
KAPPA = -19000.0 * TS32 + 19.0 / 128.0


def design_rows(b0, b1, a0):
    """fmd_api.cpp design_pll_span in float64: rows (lpf_end, I_end, dev[N1], dev[N2], dev[L-1]) over v = (lpf, I, e1, e2, r0, eh[0..L-1])."""
    NV = 5 + L
    def unit(i):
        v = np.zeros(NV); v[i] = 1.0; return v
    lpf, I, e1, e2 = unit(0), unit(1), unit(2), unit(3)
    dev = np.zeros(NV)
    rows = [None] * 5
    g0 = None
    for n in range(L):
        lpf = b0 * e2 + b1 * e1 + a0 * lpf
        I = I + KTSI * e1
        g = -100.0 * (0.01 * lpf + I)
        if n == 0:
            g0 = g.copy()
        dev = dev + TS32 * (g - g0)
        dev[4] -= TS32
        e = TWO_PI * dev
        e = e.copy(); e[5 + n] += TWO_PI
        if n == N1: rows[2] = dev.copy()
        if n == N2: rows[3] = dev.copy()
        if n == L - 1:
            rows[0] = lpf.copy(); rows[1] = I.copy(); rows[4] = dev.copy()
        e2, e1 = e1, e
    R = np.array(rows)
    w, s = R[:, 5:], R[:, :5]
    x = np.array([N1, N2, L - 1], float)
    A = np.stack([x, x ** 2, x ** 3], axis=1)
    return w, s, np.linalg.inv(A)


def clamp(x, lo=-1.0, hi=1.0):
    return min(hi, max(lo, x))


def wrap(x):
    return x - np.round(x)


def peak_rail(fm_out, k):
    """P'[n] on x' = fm_out delayed by 31 (as k_pll_span runs it), float64."""
    pk, pa0, pa1 = float(k.pilot_b[0]), float(k.pilot_a[0]), float(k.pilot_a[1])
    n = fm_out.size
    xd = np.concatenate([np.zeros(31), fm_out])[:n]
    x2 = np.concatenate([np.zeros(2), xd])
    P = np.zeros(n + 2)
    for i in range(n):
        P[i + 2] = pk * x2[i] + pa1 * P[i + 1] + pa0 * P[i]
    return P            # P[i + 2] = P'[i]


def span_update(st, eh, tab, dense_wrap=True):
    """One span of k_pll_span given the held-frequency errors eh[0..127] (turns).  Returns the cubic (c0..c3)."""
    w, s, minv = tab
    b0, b1, a0 = st["b"]
    lpf, integ, e1, e2, tb = st["lpf"], st["integ"], st["e1"], st["e2"], st["t"]
    return lpf, integ, e1, e2, tb


class Loop:
    """State and span step shared by the dense and the sparse model (k_pll_span's arithmetic, float64 except the float32 frequency word)."""

    def __init__(self, k):
        self.b0, self.b1, self.a0 = float(k.pll_lpf_b[0]), float(k.pll_lpf_b[1]), float(k.pll_lpf_a[0])
        self.w, self.s, self.minv = design_rows(self.b0, self.b1, self.a0)
        self.lpf = self.integ = self.e1 = self.e2 = self.t = 0.0

    def hold(self):
        lpf0 = self.b0 * self.e2 + self.b1 * self.e1 + self.a0 * self.lpf
        ig0 = clamp(KTSI * self.e1 + self.integ)
        u0 = lpf0 * 0.01 + ig0
        cc = clamp(u0)
        F0 = float(np.float32(np.float32(cc) * np.float32(-100.0) + np.float32(-19000.0)))     # the reference's float frequency word
        gq = F0 + 19000.0
        r0 = 100.0 * cc + gq
        dq = gq * TS32 + KAPPA
        return u0, gq, r0, dq

    def finish(self, rowv, u0, gq, r0, dq, e_last, e_prev):
        u_end = rowv[0] * 0.01 + clamp(rowv[1])
        railed = abs(u0) >= 1.0 and abs(u_end) >= 1.0 and u0 * u_end > 0.0
        d1, d2, d3 = (0.0, 0.0, 0.0) if railed else (rowv[2], rowv[3], rowv[4])
        al, be, ga = self.minv @ np.array([d1, d2, d3])
        pc = (self.t + dq, dq + al, be, ga)
        self.lpf = rowv[0]
        self.integ = clamp(rowv[1])
        g_end = -100.0 * (self.lpf * 0.01 + self.integ)
        d3m = d3 - TS32 * (g_end - gq)
        self.e1 = wrap(e_last + d3) * TWO_PI
        self.e2 = wrap(e_prev + d3m) * TWO_PI
        self.t = wrap(L * dq + self.t + d3)
        return pc

    def state_vec(self, r0):
        return np.array([self.lpf, self.integ, self.e1, self.e2, r0])


def eval_poly(pc):
    u = np.arange(L, dtype=np.float64)
    c19 = ((19 * (np.arange(L) + 1)) & 127) / 128.0
    return wrap(((pc[3] * u + pc[2]) * u + pc[1]) * u + pc[0] - c19)


def run_dense(fm_out, k, n_spans=None):
    """k_pll_span as it is (round 3), without the warm-up rail: psi per sample from the real rail and its symmetric difference."""
    P = peak_rail(fm_out, k)
    n = fm_out.size
    h = np.array(k.b_hilbert, np.float64)
    H = np.sum(h * np.exp(-1j * W0 * np.arange(65)))
    quad = abs(H) / (2 * np.sin(W0))
    re = P[1:n + 1]
    im = quad * (P[0:n] - P[2:n + 2])
    psi = np.arctan2(im, re) / TWO_PI
    lp = Loop(k)
    out = np.zeros(n)
    fa = np.arange(L) + 1.0
    c19 = ((19 * (np.arange(L) + 1)) & 127) / 128.0
    states = []
    for s0 in range(0, n, L):
        states.append((lp.lpf, lp.integ, lp.e1, lp.e2, lp.t))
        u0, gq, r0, dq = lp.hold()
        eh = wrap(fa * dq + psi[s0:s0 + L] + (lp.t - c19))
        rowv = lp.w @ eh + lp.s @ lp.state_vec(r0)
        pc = lp.finish(rowv, u0, gq, r0, dq, eh[-1], eh[-2])
        out[s0:s0 + L] = eval_poly(pc)
    return out, psi, states


def bandpass_taps(ntaps, cutoff_hz):
    """Linear-phase low-pass prototype (Kaiser window) applied to the DOWN-MIXED signal: passband around the pilot."""
    if ntaps <= 1:
        return np.ones(1)
    if cutoff_hz < 0:                        # boxcar: nulls at multiples of fs / ntaps offset from the pilot
        return np.ones(ntaps) / ntaps
    n = np.arange(ntaps) - (ntaps - 1) / 2.0
    fc = cutoff_hz / 128000.0
    hlp = 2 * fc * np.sinc(2 * fc * n) * np.kaiser(ntaps, 7.0)
    return hlp / hlp.sum()


def run_sparse(fm_out, k, D=16, pre_taps=65, cutoff_hz=3000.0, fit_deg=1, warm_states=None, warm_spans=64, verbose=False):
    """The sparse form.  Points n_k = D k + D - 1 of every span (the last one is the span's last sample)."""
    pk, pa0, pa1 = float(k.pilot_b[0]), float(k.pilot_a[0]), float(k.pilot_a[1])
    r = np.sqrt(-pa0)
    wp = np.arccos(pa1 / (2.0 * r))             # the poles the float32 coefficients really have: r exp(+-j wp), wp != w0 by ~1e-8 rad
    rho = r * np.exp(1j * (wp - W0))            # Z[m] = rho Z[m-1] + u[m]: a pole 1e-8 rad off the real axis is 1e-4 rad of pilot phase
    if verbose: print("pole angle - w0 =", wp - W0, "rad; 1 - r =", 1 - r)
    n = fm_out.size
    KP = L // D
    nk = D * np.arange(KP) + D - 1
    # --- band-limited, decimated Z: Zt[m'] for m' = nk - 32 (span relative) ---------------------------------------
    # u[m] = exp(-j w0 m) x[m-2]; prefilter h (delay dl) : ut[m] = sum_i h[i] u[m + dl - i]  (centred)
    hpre = bandpass_taps(pre_taps, cutoff_hz)
    dl = (pre_taps - 1) // 2
    m = np.arange(n + dl)
    x_m2 = np.concatenate([np.zeros(2), fm_out, np.zeros(dl)])[:n + dl]     # x[m-2]
    u = np.exp(-1j * W0 * m) * x_m2
    ut = np.convolve(u, hpre)[dl:dl + n]                                   # centred: ut[m] ~ band-limited u[m], uses u[m-dl .. m+dl]
    # exact decimated one-pole on ut at m' = (span start) + nk - 32
    Zfull = np.zeros(n, complex)
    z = 0j
    for i in range(n):
        z = rho * z + ut[i]
        Zfull[i] = z
    # arg A = arg(-j (K / sin w0) e^{j w0} Z)
    cA = -1j * (pk / np.sin(wp)) * np.exp(1j * wp)
    # fit basis: eh_fit[n] = sum_k ell[n][k] eh_k
    V = np.vander((nk - 63.5) / 64.0, fit_deg + 1, increasing=True)
    Vall = np.vander((np.arange(L) - 63.5) / 64.0, fit_deg + 1, increasing=True)
    ell = Vall @ np.linalg.pinv(V)                                      # [L][KP]
    lp = Loop(k)
    wS = lp.w @ ell                                                     # [5][KP]
    sufw = np.cumsum(lp.w[:, ::-1], axis=1)[:, ::-1]                    # sufw[r][n] = sum_{n' >= n} w[r][n']
    out = np.zeros(n)
    n_wraps = 0
    for si, s0 in enumerate(range(0, n, L)):
        if warm_states is not None and si < warm_spans:
            lp.lpf, lp.integ, lp.e1, lp.e2, lp.t = warm_states[si + 1] if si + 1 < len(warm_states) else warm_states[si]
            continue
        u0, gq, r0, dq = lp.hold()
        mp = s0 + nk - 32
        Zk = np.where(mp >= 0, Zfull[np.maximum(mp, 0)], 0)
        A = cA * Zk
        # smooth psi[n] - c19[n] = arg A[m-32] + w0 (m - 32) - 19 (n+1)/128 (turns): 19 * 128 s / 128 is whole, so
        # = arg(A)/2pi + 19 (n - 32)/128 - 19 (n+1)/128 = arg(A)/2pi - 19*33/128
        phi = np.angle(A) / TWO_PI - 19.0 * 33.0 / 128.0 + (W0 * (s0 % 128)) * 0
        ek = wrap(phi + lp.t + (nk + 1.0) * dq)
        # unwrap along the points
        ek = ek[0] + np.concatenate([[0.0], np.cumsum(wrap(np.diff(ek)))])
        fit = ell @ ek
        rowv = wS @ ek + lp.s @ lp.state_vec(r0)
        rnd = np.round(fit)
        if np.any(rnd != 0):
            n_wraps += 1
            rowv = rowv - lp.w @ rnd                                    # (the kernel: suffix sums of the rows at the crossing)
        pc = lp.finish(rowv, u0, gq, r0, dq, wrap(fit[-1]), wrap(fit[-2]))
        out[s0:s0 + L] = eval_poly(pc)
    return out, n_wraps


def rms(x):
    return float(np.sqrt(np.mean(np.square(x))))


if __name__ == "__main__":
    fs, bs = 256000, 16384
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    def stress_capture(n, tones, seed, u8=False):
        """audio content right where it aliases onto the pilot at 8 kHz point spacing (11 kHz = 19 - 8) and at the band edge (15 kHz)"""
        rng = np.random.default_rng(seed)
        t = np.arange(n) / fs
        left = sum(a * np.sin(TWO_PI * f * t + 0.3 * i) for i, (f, a) in enumerate(tones))
        right = 0.3 * np.sin(TWO_PI * 700.0 * t)
        p = TWO_PI * 19000.0 * t
        mpx = 0.4 * (left + right) / 1.6 + 0.1 * np.sin(p) + 0.4 * (left - right) / 1.6 * np.sin(2 * p)
        iq = np.exp(1j * TWO_PI * 75000.0 * np.cumsum(mpx) / fs) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        return synth.u8_to_cf32(synth.to_u8(iq)) if u8 else synth.to_cf32(iq)
    cases = [(11, {}), (13, {"noise_sigma": 0.1}), (21, "stress11k"), (22, "stress15k"), (23, "stress_u8")]
    for seed, kw in cases:
        if isinstance(kw, str):
            tones = {"stress11k": [(11000.0, 0.6), (3000.0, 0.2)], "stress15k": [(15000.0, 0.6), (14500.0, 0.3)], "stress_u8": [(11000.0, 0.5), (8000.0, 0.4)]}[kw]
            cap = stress_capture(bs * nb, tones, seed, u8=(kw == "stress_u8"))
        else:
            cap = synth.to_cf32(synth.fm_capture(bs * nb, fs=float(fs), seed=seed, **kw)["iq"])
        o = O.run_chain(cap, bs, fs, u8=False, streams=["fm_out", "pll_dt"])
        k = o["coeffs"]
        fo = o["fm_out"].astype(np.float64)
        ref = o["pll_dt"].astype(np.float64)
        dense, psi, states = run_dense(fo, k)
        def show(name, t):
            d = wrap(t - ref)
            blk = [rms(d[i * 8192:(i + 1) * 8192]) for i in range(d.size // 8192)]
            print(f"  {name:34s}", " ".join(f"{x:.1e}" for x in blk))
        print(f"seed {seed} {kw}: NCO phase error vs oracle, RMS per block (turns)")
        show("dense (round 3, no warm-up rail)", dense)
        for (D, pt, fc, deg) in [(16, 1, 0, 1), (16, 17, -1, 1), (16, 33, 3000, 1), (16, 65, 3000, 1), (32, 1, 0, 1), (32, 33, -1, 1)]:
            sp, nw = run_sparse(fo, k, D=D, pre_taps=pt, cutoff_hz=fc or 3000.0, fit_deg=deg, warm_states=states, warm_spans=64)
            show(f"sparse D={D} pre={pt} fc={fc} deg={deg} wraps={nw}", sp)

# ---- how long must the dense form run after a reset? (python tools/proto/sparse_pll.py 10 warm) ----
if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "warm":
    print("=== warm-up length ===")
    for seed, kw in [(11, {}), (13, {"noise_sigma": 0.1}), (31, {"pilot_hz": 19005.0})]:
        cap = synth.to_cf32(synth.fm_capture(bs * nb, fs=float(fs), seed=seed, **kw)["iq"])
        o = O.run_chain(cap, bs, fs, u8=False, streams=["fm_out", "pll_dt"])
        k = o["coeffs"]; fo = o["fm_out"].astype(np.float64); ref = o["pll_dt"].astype(np.float64)
        dense, psi, states = run_dense(fo, k)
        for ws in (0, 1, 2, 4, 16, 64):
            sp, nw = run_sparse(fo, k, D=16, pre_taps=17, cutoff_hz=-1, fit_deg=1, warm_states=states, warm_spans=ws)
            d = wrap(sp - ref)
            blk = [rms(d[max(i * 8192, ws * 128):(i + 1) * 8192]) for i in range(d.size // 8192)]
            print(f"seed {seed} warm_spans={ws:3d} wraps={nw}:", " ".join(f"{x:.1e}" for x in blk))


# ---------------------------------------------------------------------------------------------------------------------------
# The same thing organised as the kernel does it (k_pll_sparse): lane = point, 16 samples per lane, new / old half sums, rotation,
# scan with rho^16, one arctangent per point, linear fit through the 8 points.  float32 where the kernel is float32.
# ---------------------------------------------------------------------------------------------------------------------------
def design_sparse(k):
    pk, pa0, pa1 = float(k.pilot_b[0]), float(k.pilot_a[0]), float(k.pilot_a[1])
    r = np.sqrt(-pa0)
    wp = np.arccos(pa1 / (2.0 * r))
    rho = r * np.exp(1j * (wp - W0))
    # W[q], q = -8 .. 23: 17-tap boxcar in front of the exact decimation by 16
    W = np.zeros(32, complex)
    for q in range(-8, 24):
        W[q + 8] = sum(rho ** i for i in range(16) if abs(q - i) <= 8) / 17.0
    t = np.arange(32)
    # points at n_k = 16 k + 9: a point's 32 inputs are then exactly two 16-sample columns of fm_out (k - 3: "old" half, k - 2: "new" half),
    # which the front end's tiles can sum on the matrix cores (m'_k = span + 16 k - 23)
    Wc = W[(23 - t) + 8] * np.exp(-1j * W0 * (t - 46))          # tap t multiplies x[m' - 25 + t]
    rot = np.exp(-1j * W0 * 16 * np.arange(8))
    cA = -1j * (pk / np.sin(wp)) * np.exp(1j * wp)
    phi0 = np.angle(cA) / TWO_PI - 19.0 * 33.0 / 128.0
    phi0 -= np.round(phi0)
    lp_b = (float(k.pll_lpf_b[0]), float(k.pll_lpf_b[1]), float(k.pll_lpf_a[0]))
    w, s, minv = design_rows(*lp_b)
    nk = 16 * np.arange(8) + 9
    nbar = nk.mean()                                             # 65
    nn = np.arange(L)
    h = np.array(k.b_hilbert, np.float64)
    g = abs(np.sum(h * np.exp(-1j * W0 * np.arange(65))))
    kap2 = -np.exp(-2j * wp) / (W.sum() * (1 - r * np.exp(-1j * (wp + W0))))     # the filter's second, non-resonant pole branch
    return {
        "kap2": kap2, "Wc": Wc, "rot": rot, "rho16": rho ** 16, "phi0": phi0, "nk": nk, "nbar": nbar,
        "inv_s2": 1.0 / np.sum((nk - nbar) ** 2),
        "wsum": w.sum(axis=1), "wmom": (w * (nn - nbar)).sum(axis=1), "s": s, "minv": minv,
        "sw": np.concatenate([np.cumsum(w[:, ::-1], axis=1)[:, ::-1], np.zeros((5, 1))], axis=1),
        "pw_scale": abs(cA) ** 2 * (1 + g * g) / 2 * 16.0, "b": lp_b,
    }


def run_sparse_kernel_like(fm_out, k, start_span=1, start_state=None, f32=True):
    T = design_sparse(k)
    F = np.float32 if f32 else np.float64
    Cx = np.complex64 if f32 else np.complex128
    n = fm_out.size
    x = np.concatenate([np.zeros(192), fm_out]).astype(F)        # history pad in front (kFoPad)
    Wc = T["Wc"].astype(Cx); rot = T["rot"].astype(Cx); rho16 = Cx(T["rho16"])
    lp = Loop(k)
    out = np.zeros(n)
    Z = Cx(0)
    O_prev = None
    power = 0.0
    for q in range(n // L):
        base = 192 + L * q
        N = np.zeros(8, Cx); Oh = np.zeros(8, Cx)
        for kk in range(8):
            c = x[base + 16 * kk - 32: base + 16 * kk - 16]      # column k - 2
            N[kk] = np.sum(Wc[16:] * c, dtype=Cx); Oh[kk] = np.sum(Wc[:16] * c, dtype=Cx)
        if O_prev is None:                                       # kernel start: the old half over the column in front
            c = x[base - 48: base - 32]
            O_prev = np.sum(Wc[:16] * c, dtype=Cx)
        V = rot * (N + np.concatenate([[O_prev], Oh[:7]]))
        O_prev = Oh[7]
        Zk = np.zeros(8, Cx)
        for kk in range(8):
            Z = Cx(rho16 * Z + V[kk]); Zk[kk] = Z
        Zk = (Zk + Cx(T["kap2"]) * V).astype(Cx)
        power += float(np.sum(np.abs(Zk) ** 2)) * T["pw_scale"]
        if q < start_span:
            if start_state is not None and q == start_span - 1:
                lp.lpf, lp.integ, lp.e1, lp.e2, lp.t = start_state
            continue
        u0, gq, r0, dq = lp.hold()
        phi = (np.angle(Zk.astype(np.complex128)) / TWO_PI + T["phi0"]).astype(F)
        ek = wrap((phi + F(lp.t) + (T["nk"] + 1.0).astype(F) * F(dq)).astype(np.float64))
        ek = ek[0] + wrap(ek - ek[0])
        a = ek.sum() / 8.0
        b = ((T["nk"] - T["nbar"]) * ek).sum() * T["inv_s2"]
        rowv = T["wsum"] * a + T["wmom"] * b + T["s"] @ lp.state_vec(r0)
        f0, f127 = a - T["nbar"] * b, a + (127 - T["nbar"]) * b
        r_lo, r_hi = np.round(f0), np.round(f127)
        if r_lo != 0 or r_hi != 0:
            corr = r_lo * T["sw"][:, 0]
            if r_hi != r_lo:
                fit = a + b * (np.arange(L) - T["nbar"])
                nstar = int(np.argmax(np.round(fit) == r_hi))
                corr = corr + (r_hi - r_lo) * T["sw"][:, nstar]
            rowv = rowv - corr
        pc = lp.finish(rowv, u0, gq, r0, dq, wrap(f127), wrap(f127 - b))
        out[L * q: L * q + L] = eval_poly(pc)
    return out, power


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "kernel":
    print("=== kernel-like organisation (float32) vs the float64 model ===")
    for seed, kw in [(11, {}), (13, {"noise_sigma": 0.1}), (12, {"pilot_hz": 19040.0})]:
        cap = synth.to_cf32(synth.fm_capture(bs * nb, fs=float(fs), seed=seed, **kw)["iq"])
        o = O.run_chain(cap, bs, fs, u8=False, streams=["fm_out", "pll_dt", "pilot"])
        k = o["coeffs"]; fo = o["fm_out"].astype(np.float64); ref = o["pll_dt"].astype(np.float64)
        dense, psi, states = run_dense(fo, k)
        sp, nw = run_sparse(fo, k, D=16, pre_taps=17, cutoff_hz=-1, fit_deg=1, warm_states=states, warm_spans=64)
        for f32 in (False, True):
            kl, power = run_sparse_kernel_like(fo, k, start_span=64, start_state=states[64], f32=f32)
            d = wrap(kl - ref); d2 = wrap(kl - sp)
            pil = o["pilot"].astype(np.float64).reshape(-1, 2)
            print(f"seed {seed} f32={f32}: vs oracle", " ".join(f"{rms(d[i * 8192:(i + 1) * 8192]):.1e}" for i in range(1, d.size // 8192)),
                  "| vs float64 model", f"{rms(d2[8192:]):.1e}", "| wraps", nw, "| power ratio", power / np.sum(pil ** 2))
