"""Offline model (float64) of the round-3 tolerance-mode pilot PLL against the CPU oracle (development tool; tests/-style use of oracle/).

  * the pilot peak filter runs on the REAL rail only (fm_out delayed by 31 samples); the quadrature rail of the filtered pilot is
    the symmetric difference q (P[n-2] - P[n]) around the real rail P[n-1]  (the filter has real coefficients and commutes with
    the Hilbert FIR; behind the filter the signal is a 4 Hz wide line at 19 kHz, where the Hilbert FIR is a gain and a quarter turn)
  * psi[n] = arg(pilot[n]) once per sample, independent of the loop
  * the loop is advanced one span (L samples) at a time: the error sequence under a held NCO frequency is
    e_hold[n] = wrap(psi[n] + 2 pi (t0 + (n+1) F0 Ts)); the loop filter / integrator / NCO are linear, so the state after the span
    and the NCO phase at chosen samples are fixed weight vectors (designed in double, feedback of the phase deviation included
    exactly) applied to e_hold
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import oraclelib as O
import synth

TWO_PI = 2 * np.pi
TS = 1.0 / 128000.0
KTSI = 0.1 * TS


def wrap_pi(x):
    return x - TWO_PI * np.round(x / TWO_PI)


def design_span(L, b0, b1, a0):
    """Linear map of one span under the hold.  Conventions (reference broadcast_fm_demod.cpp:430-456): at sample n the loop uses
    e[n-1]:  lpf[n] = b0 x1 + b1 e[n-1] + a0 lpf[n-1] (x1 = e[n-2]),  I[n] = I[n-1] + ktsi e[n-1],  f[n] = -19000 - 100 (0.01 lpf[n] + I[n]),
    t[n] = t[n-1] + Ts f[n],  e[n] = e_hold[n] + 2 pi d[n],  d[n] = t[n] - (t[-1] + (n+1) Ts f[0]).
    Unknowns are linear in v = (lpf[-1], I[-1], e[-1], e[-2], e_hold[0..L-1]).  Returns matrices giving e[0..L-1], lpf[L-1], I[L-1],
    d[0..L-1] as functions of v (the constant -19000 cancels in d)."""
    nv = 4 + L
    def unit(i):
        v = np.zeros(nv); v[i] = 1.0; return v
    # propagate symbolic (as coefficient vectors over v)
    lpf = unit(0); I = unit(1); e1 = unit(2); e2 = unit(3)
    E = []; D = []; LPF = []; INT = []
    d = np.zeros(nv); g0 = None
    for n in range(L):
        lpf = b0 * e2 + b1 * e1 + a0 * lpf
        I = I + KTSI * e1
        g = -100.0 * (0.01 * lpf + I)          # f[n] + 19000
        if n == 0:
            g0 = g.copy()
        d = d + TS * (g - g0)
        e = unit(4 + n) + TWO_PI * d
        E.append(e); D.append(d.copy()); LPF.append(lpf.copy()); INT.append(I.copy())
        e2 = e1; e1 = e
    return np.array(E), np.array(D), np.array(LPF), np.array(INT), g0


def run_model(fm_out, k, L=64, quad_mode="sym", verbose=False):
    """fm_out: float64 [n] (the oracle's fm_out stream, concatenated blocks).  Returns t[n] (NCO phase, wrapped) per sample."""
    n = fm_out.size
    pk, pa0, pa1 = float(k.pilot_b[0]), float(k.pilot_a[0]), float(k.pilot_a[1])
    b0, b1, a0 = float(k.pll_lpf_b[0]), float(k.pll_lpf_b[1]), float(k.pll_lpf_a[0])
    # Hilbert FIR gain at 19 kHz
    h = np.array(k.b_hilbert, np.float64)
    w0 = TWO_PI * 19000.0 / 128000.0
    H = np.sum(h * np.exp(-1j * w0 * np.arange(65)))
    # H = g * exp(-j (w0*32 + pi/2))?  the reference stores taps reversed; measure gain and check the phase
    g = abs(H)
    q = g / (2 * np.sin(w0))
    # IIR on the stream delayed by 31: P'[n] = K x'[n-2] + a1 P'[n-1] + a0 P'[n-2], x'[n] = fm_out[n-31]
    xd = np.concatenate([np.zeros(31), fm_out])[:n]
    P = np.zeros(n + 2)   # P[i+2] = P'[i]
    x2 = np.concatenate([np.zeros(2), xd])
    for i in range(n):
        P[i + 2] = pk * x2[i] + pa1 * P[i + 1] + pa0 * P[i]
    re = P[1:n + 1]                    # P'[n-1]
    sign = np.sign(np.imag(H * np.exp(1j * w0 * 32)))   # +1: H = +j g e^{-j 32 w} (cos -> -sin)
    im = sign * q * (P[0:n] - P[2:n + 2])   # q (Pd[n-1] - Pd[n+1])
    psi = np.arctan2(im, re)
    E, D, LPF, INT, g0 = design_span(L, b0, b1, a0)
    t = np.zeros(n)
    lpf, I, e1, e2, tprev = 0.0, 0.0, 0.0, 0.0, 0.0
    for s0 in range(0, n, L):
        v = np.zeros(4 + L)
        v[0], v[1], v[2], v[3] = lpf, I, e1, e2
        F0 = -19000.0 + float(g0[:4] @ v[:4])
        th = tprev + (np.arange(L) + 1) * F0 * TS
        eh = wrap_pi(psi[s0:s0 + L] + TWO_PI * th)
        v[4:] = eh
        e = E @ v
        d = D @ v
        tt = th + d
        t[s0:s0 + L] = tt - np.round(tt)
        lpf = float(LPF[-1] @ v); I = float(INT[-1] @ v)
        e1 = float(wrap_pi(e[-1])); e2 = float(wrap_pi(e[-2]))
        tprev = float(t[s0 + L - 1])
    return t, psi, (re, im)


if __name__ == "__main__":
    fs = 256000
    bs = 16384
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    for seed, kw in [(11, {}), (12, {"pilot_hz": 19040.0}), (13, {"noise_sigma": 0.1})]:
        cap = synth.to_cf32(synth.fm_capture(bs * nb, fs=float(fs), seed=seed, **kw)["iq"])
        o = O.run_chain(cap, bs, fs, u8=False, streams=["fm_out", "fm_out_iq", "pilot", "pll_dt"])
        k = o["coeffs"]
        fo = o["fm_out"].astype(np.float64)
        for L in (64,):
            t, psi, (re, im) = run_model(fo, k, L)
            ref = o["pll_dt"].astype(np.float64)
            dlt = t - ref; dlt -= np.round(dlt)
            pil = o["pilot"].astype(np.float64).reshape(-1, 2)
            n = ref.size
            blkrms = [np.sqrt(np.mean(dlt[i * 8192:(i + 1) * 8192] ** 2)) for i in range(n // 8192)]
            amp = np.sqrt(np.mean(pil[-8192:, 0] ** 2 + pil[-8192:, 1] ** 2))
            rail = np.sqrt(np.mean((re[-8192:] - pil[-8192:, 0]) ** 2 + (im[-8192:] - pil[-8192:, 1]) ** 2)) / amp
            print(f"seed {seed} {kw} L={L}: phase err rms per block (turns):", " ".join(f"{x:.1e}" for x in blkrms), f"| rail mismatch {rail:.1e}")
