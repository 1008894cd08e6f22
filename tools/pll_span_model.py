"""Float64 model of k_pll_fast's span-held PLL evaluation against the serial loop (development tool, numpy only): the NCO frequency
word of a span's first sample is held for S samples, the loop filter then yields every sample's own word and phases / phase errors
are corrected to first order.  Prints the RMS phase difference to the serial loop for S = 8 ... 128 (5e-8 turns at S = 64)."""
import numpy as np
rng = np.random.default_rng(1)
fs = 128000.0; Ts = 1/fs; N = 128000
n = np.arange(N)
pilot = 0.1*np.exp(1j*(2*np.pi*19000.2*n/fs + 0.3)) + 0.02*(rng.standard_normal(N)+1j*rng.standard_normal(N))
# loop coefficients (100 Hz lpf bilinear)
k = 100.0/(fs/2); t = np.tan(k*np.pi/2); two_a = 1/t; B0 = two_a+1
b0 = 1/B0; b1 = b0; a0 = -((1-two_a)/B0)
ktsi = 0.1*Ts
def clamp(x): return max(-1.0, min(1.0, x))
def det(x, t):
    return np.angle(x*np.exp(2j*np.pi*t))   # e = arg(pilot) + 2 pi t
def serial():
    lx1=ly1=integ=err=0.0; tph=0.0; out=np.zeros(N)
    for i in range(N):
        y1 = lx1*b0 + ly1*a0 + err*b1
        ig = clamp(err*ktsi + integ)
        F = -19000.0 - 100.0*clamp(0.01*y1 + ig)
        tph = tph + F*Ts; tph -= np.round(tph)
        e = det(pilot[i], tph)
        out[i] = tph
        lx1 = err; ly1 = y1; integ = ig; err = e
    return out
def span(S, corr_feedback=False):
    lx1=ly1=integ=err=0.0; tph=0.0; out=np.zeros(N); pos=0
    while pos < N:
        m = min(S, N-pos)
        y1_0 = lx1*b0 + ly1*a0 + err*b1
        ig_0 = clamp(err*ktsi + integ)
        F = -19000.0 - 100.0*clamp(0.01*y1_0 + ig_0)
        t = tph + (np.arange(m)+1)*F*Ts; t -= np.round(t)
        e = det(pilot[pos:pos+m], t)
        # filter over the span with uncorrected e
        ee = np.concatenate([[lx1, err], e])   # e_{-2}, e_{-1}, e_0..
        y1 = np.zeros(m); ig = np.zeros(m)
        y1[0] = y1_0; ig[0] = ig_0
        for j in range(1, m):
            y1[j] = a0*y1[j-1] + b0*ee[j] + b1*ee[j+1]
            ig[j] = ig[j-1] + ktsi*ee[j+1]
        Fj = -19000.0 - 100.0*np.clip(0.01*y1 + ig, -1, 1)
        d = (Fj - F)*Ts; d[0] = 0
        c = np.cumsum(d)
        tc = t + c
        ec = e + 2*np.pi*c
        ec = np.where(ec > np.pi, ec-2*np.pi, ec); ec = np.where(ec < -np.pi, ec+2*np.pi, ec)
        out[pos:pos+m] = tc
        tph = tc[-1]; ly1 = y1[-1]; integ = ig[-1]; err = ec[-1]; lx1 = ec[-2] if m > 1 else err_prev_keep(lx1)
        pos += m
    return out
def err_prev_keep(x): return x
ref = serial()
for S in (8, 16, 32, 64, 128):
    o = span(S)
    d = o - ref; d -= np.round(d)
    print(S, "rms err turns", np.sqrt(np.mean(d[N//2:]**2)), "max", np.abs(d[N//2:]).max())
