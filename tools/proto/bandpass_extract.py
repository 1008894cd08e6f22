#!/usr/bin/env python3
"""Round-5 prototype (float64, CPU, test infrastructure): the L-R and RDS rails as fixed complex band-pass decimating FIRs of fm_out followed
by ONE rotation per output by the pilot NCO's slow phase deviation — against the reference's order (mix every 128 kHz sample with the NCO's
2nd / 3rd harmonic, then low-pass and decimate: broadcast_fm_demod.cpp:463-536, apply_harmonic_pll.cpp:88-139).

  reference:  y[m] = sum_t h[t - 4m'] a[t] e^{j 2 pi (H dt[t] + off)},   a[t] = x[t-32] + j hilbert(x)[t]
  here:       dt[t] = -(19/128)(t+1) + phi[t]  (phi: the loop's slow deviation)  =>
              y[m] ~= e^{j 2 pi (H phi(t_c) + off)} sum_u G_H[u - ...] x[u],      G_H = (h e^{-j 2 pi H 19/128 .}) * (delta_32 + j hilbert)
  exact when phi is constant over the FIR's 128 taps (1 ms); otherwise the filter is seen shifted by H phi' (Hz).

Prints the RMS difference of both forms against the oracle's lmr / rds streams per block."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "tests"), str(ROOT / "oracle")]
import oraclelib as O
import synth


FIRST_ORDER = False


def run(fs=256000, bs=16384, nb=24, seed=11, channel=1, **cap_kw):
    cap = synth.to_cf32(synth.fm_capture(bs * nb, fs=float(fs), seed=seed, channel=channel, **cap_kw)["iq"])
    o = O.run_chain(cap, bs, fs, u8=False, streams=["fm_out", "pll_dt", "lmr", "lpr", "rds", "lmr_phase", "fm_out_iq", "audio"])
    k = o["coeffs"]
    x = o["fm_out"].astype(np.float64)
    dt = o["pll_dt"].astype(np.float64)
    n = x.size
    nfo = n // nb
    hil = k.arr("b_hilbert").astype(np.float64)          # 65 taps, y[i] = sum_n s[i - 64 + n] b[n]
    h_lmr = k.arr("b_lmr").astype(np.float64)
    h_rds = k.arr("b_rds").astype(np.float64)
    xp = np.concatenate([np.zeros(64), x])
    im = np.array([np.dot(xp[i:i + 65], hil) for i in range(n)])
    re = np.concatenate([np.zeros(32), x])[:n]
    a = re + 1j * im
    iq = o["fm_out_iq"].astype(np.float64).reshape(-1, 2)
    print("analytic signal vs oracle fm_out_iq: max", np.abs(a - (iq[:, 0] + 1j * iq[:, 1])).max())
    # per-block L-R offset: lmr_phase[b] = the accumulator AFTER block b; block b is mixed with the value after block b-1 (0 for b = 0)
    lp = o["lmr_phase"].astype(np.float64)
    off = np.concatenate([[0.0], lp[:-1]]) if lp.size == nb else np.zeros(nb)
    off_t = np.repeat(off, nfo)
    res = {}
    for name, H, taps, M, ostream in (("lmr", 2.0, h_lmr, 4, "lmr"), ("rds", 3.0, h_rds, 8, "rds")):
        NN = taps.size
        # exact (reference order), float64
        mix = a * np.exp(2j * np.pi * (H * dt + (off_t if name == "lmr" else 0.0)))
        mp = np.concatenate([np.zeros(NN, complex), mix])
        n_out = n // M
        idx = (np.arange(n_out) + 1) * M            # window = samples [M(i+1) - NN, M(i+1)) of the stream
        ex = np.array([np.dot(mp[i:i + NN], taps) for i in idx])
        # band-pass form: nominal carrier n[t] = -H 19/128 (t + 1) turns; deviation = H dt - n (slow), evaluated at the window's centre
        t = np.arange(n, dtype=np.float64)
        nom = -H * 19.0 / 128.0 * (t + 1.0)
        dev = H * dt - nom
        dev = dev - np.round(dev)
        dev = np.unwrap(dev * 2 * np.pi) / (2 * np.pi)
        bp = a * np.exp(2j * np.pi * nom)
        bpp = np.concatenate([np.zeros(NN, complex), bp])
        s = np.array([np.dot(bpp[i:i + NN], taps) for i in idx])
        devp = np.concatenate([np.full(NN, dev[0]), dev])
        c0 = idx + NN // 2 - 1                      # the two centre samples of the window in the padded array
        dev_c = 0.5 * (devp[c0] + devp[c0 + 1])
        # first-order term: the deviation's slope over the window (turns / sample) times the FIR with taps (tau - 63.5) h[tau]
        slope = devp[c0 + 1] - devp[c0]
        s1 = np.array([np.dot(bpp[i:i + NN], taps * (np.arange(NN) - (NN - 1) / 2)) for i in idx])
        if FIRST_ORDER:
            s = s + 2j * np.pi * slope * s1
        if name == "lmr":
            # block edges: a window that straddles two blocks mixes its older samples with the older offset (reference: the FIR's history)
            offp = np.concatenate([np.full(NN, off_t[0]), off_t])
            ap = np.empty(n_out, complex)
            for j, i in enumerate(idx):
                o_w = offp[i:i + NN]
                if o_w[0] == o_w[-1]:
                    ap[j] = s[j] * np.exp(2j * np.pi * (dev_c[j] + o_w[-1]))
                else:
                    cut = np.argmax(o_w != o_w[0])
                    s_old = np.dot(bpp[i:i + cut], taps[:cut])
                    ap[j] = np.exp(2j * np.pi * dev_c[j]) * (s_old * np.exp(2j * np.pi * o_w[0]) + (s[j] - s_old) * np.exp(2j * np.pi * o_w[-1]))
        else:
            ap = s * np.exp(2j * np.pi * dev_c)
        res[name] = (ex, ap)
        if name == "lmr":
            ref = o["lmr"].astype(np.float64)
            e1, e2 = ex.imag - ref, ap.imag - ref
        else:
            ex_, ap_ = ex, ap
            e1 = e2 = None
        nblk = n_out // nb
        print(f"--- {name}: per block RMS  [exact-f64 vs oracle]  [band-pass vs oracle]  [band-pass vs exact-f64]   signal rms")
        for b in range(nb):
            sl = slice(b * nblk, (b + 1) * nblk)
            d3 = np.sqrt(np.mean(np.abs(ap[sl] - ex[sl]) ** 2))
            if name == "lmr":
                print(f"  blk {b:2d}  {np.sqrt(np.mean(e1[sl] ** 2)):.2e}  {np.sqrt(np.mean(e2[sl] ** 2)):.2e}  {d3:.2e}   {np.sqrt(np.mean(ref[sl] ** 2)):.3f}   dev slope {(dev[(b + 1) * nfo - 1] - dev[b * nfo]) / nfo * 128000 / H:+.2f} Hz")
            else:
                print(f"  blk {b:2d}  {d3:.2e}   |rds| {np.sqrt(np.mean(np.abs(ex[sl]) ** 2)):.4f}  rel {d3 / max(np.sqrt(np.mean(np.abs(ex[sl]) ** 2)), 1e-12):.2e}")
    return res


if __name__ == "__main__":
    kw = {}
    for a in sys.argv[1:]:
        if a == "first_order":
            FIRST_ORDER = True
            continue
        k, v = a.split("=")
        kw[k] = float(v) if "." in v else int(v)
    run(**kw)
