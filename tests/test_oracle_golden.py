"""The oracle (oracle/fm_oracle.c) against golden vectors dumped from the compiled reference.

These run everywhere (CPU only, no /root/reference needed): tests/golden/*.npz were produced by
tests/golden/make_golden.py from oracle/_ref/fm_ref_dump, i.e. by the reference's own sources built
with its gcc preset flags.  Integer / byte streams must be equal; float streams are required to be
BIT-IDENTICAL (the oracle restates the exact operation order of that build) except where a filter
coefficient passes through the build's rsqrtss approximation, which is CPU-specific: there the
north-star tolerance (1e-4 RMS) applies and bit equality is reported when it holds.
"""
import hashlib

import numpy as np
import pytest

import oraclelib as O
import synth
from conftest import bits_equal, describe_diff, rms
from rds_groups import decode_groups

FLOAT_STREAMS = ["fm_out_iq", "pilot", "pll", "pll_raw_err", "pll_pi_err", "lpr", "lmr", "lmr_phase", "rds",
                 "rds_raw_sym", "rds_sym", "audio", "bpsk_pll_sym", "bpsk_intdump", "bpsk_ted_raw", "bpsk_ted_pi",
                 "bpsk_pll_raw", "bpsk_pll_pi"]
TOL_RMS = 1e-4  # BASELINE.json north_star: audio / L-R / RDS within 1e-4 RMS of the reference


def _check_chain(g, out):
    exact = True
    for k in FLOAT_STREAMS:
        a, b = g[k], out[k]
        assert a.shape == b.shape, f"{k}: {describe_diff(a, b)}"
        if not bits_equal(a, b):
            exact = False
            assert rms(a - b) <= TOL_RMS * max(1.0, rms(a)), f"{k}: {describe_diff(a, b)}"
    assert np.array_equal(g["rds_count"], out["rds_count"])
    assert np.array_equal(g["rds_bytes"], out["rds_bytes"])
    return exact


def test_chain_u8_block16384(golden):
    g = golden("chain_b16384.npz")
    out = O.run_chain(g["capture"], 16384, u8=True)
    exact = _check_chain(g, out)
    # on the CPU family the fixtures were made on this is bit-exact; elsewhere rsqrtss may differ in the last bit
    if not exact:
        pytest.xfail("within tolerance but not bit-identical (rsqrtss differs on this CPU)")


def test_chain_cf32_block8192(golden):
    g = golden("chain_cf32_b8192.npz")
    out = O.run_chain(g["capture"], 8192, u8=False)
    if not _check_chain(g, out):
        pytest.xfail("within tolerance but not bit-identical (rsqrtss differs on this CPU)")


def test_long_run_rds_known_answer(golden):
    g = golden("long_b65536.npz")
    nb, bs, seed = int(g["n_blocks"]), int(g["block_size"]), int(g["seed"])
    cap_meta = synth.fm_capture(nb * bs, seed=seed)
    cap = synth.to_u8(cap_meta["iq"])
    if hashlib.sha256(cap.tobytes()).hexdigest() != str(g["capture_sha256"]):
        pytest.skip("synthetic capture not bit-reproducible with this numpy build")
    out = O.run_chain(cap, bs, u8=True, streams=["audio", "rds_sym", "lmr_phase"])
    # RDS: symbol counts, Manchester bytes identical; symbols within tolerance (bit-identical here)
    assert np.array_equal(out["rds_count"], g["rds_count"])
    assert np.array_equal(out["rds_bytes"], g["rds_bytes"])
    assert rms(out["rds_sym"] - g["rds_sym"]) <= TOL_RMS
    audio = out["audio"].reshape(nb, -1)
    for i, b in enumerate(g["audio_blocks"]):
        assert rms(audio[int(b)] - g["audio"][i]) <= TOL_RMS
    # known answer: the groups we synthesised come back out of the reference-equivalent bit stream
    want = {tuple(int(v) for v in w) for w in g["groups"]}
    got = decode_groups(out["rds_bytes"])
    assert len(got) >= 20, f"only {len(got)} RDS groups decoded"
    assert sum(1 for w in got if w in want) >= len(got) - 1
    assert all(w[0] == 0x1234 for w in got)
    if hashlib.sha256(out["audio"].tobytes()).hexdigest() != str(g["audio_sha256"]):
        pytest.xfail("audio within tolerance but not bit-identical on this CPU")


TAP_LAYOUT = [("fm_in_4", 64), ("fm_out", 64), ("fm_in_8", 64), ("lpr", 128), ("rds", 128), ("hilbert", 65),
              ("peak_b", 3), ("peak_a", 3), ("pll_b", 2), ("pll_a", 2), ("ted_b", 2), ("ted_a", 2), ("bpsk_b", 2), ("bpsk_a", 2),
              ("de1_b", 2), ("de1_a", 2), ("de50_b", 2), ("de50_a", 2), ("de75_b", 2), ("de75_a", 2), ("lpf_001", 128), ("lpf_099", 128)]


def _split_taps(flat):
    out, pos = {}, 0
    for name, n in TAP_LAYOUT:
        out[name] = flat[pos:pos + n]
        pos += n
    assert pos == flat.size
    return out


def test_designed_coefficients(golden):
    t = _split_taps(golden("taps.npz")["taps"])
    k1 = O.design(1_024_000)
    k2 = O.design(2_048_000)
    c = O.default_controls()
    exact = {
        "fm_in_4": k1.arr("b_fm_in"), "fm_out": k1.arr("b_fm_out"), "fm_in_8": k2.arr("b_fm_in"), "lpr": k1.arr("b_lpr"),
        "rds": k1.arr("b_rds"), "hilbert": k1.arr("b_hilbert"), "peak_a": k1.arr("pilot_a"),
        "pll_b": k1.arr("pll_lpf_b"), "pll_a": k1.arr("pll_lpf_a"), "ted_b": k1.arr("ted_lpf_b"), "ted_a": k1.arr("ted_lpf_a"),
        "bpsk_b": k1.arr("bpsk_lpf_b"), "bpsk_a": k1.arr("bpsk_lpf_a"), "de1_b": k1.arr("deemph_b"), "de1_a": k1.arr("deemph_a"),
    }
    for name, v in exact.items():
        assert bits_equal(t[name], v), f"{name}: {describe_diff(t[name], v)}"
    for tus in (50, 75):
        c.deemphasis_tus = tus
        k = O.design(1_024_000, c)
        assert bits_equal(t[f"de{tus}_b"], k.arr("deemph_b")) and bits_equal(t[f"de{tus}_a"], k.arr("deemph_a"))
    for hz, name in ((1, "lpf_001"), (100000, "lpf_099")):  # cut-off clamp to [0.01, 0.99]
        c.lpr_cutoff_hz = hz
        assert bits_equal(t[name], O.design(1_024_000, c).arr("b_lpr"))
    # pilot peak gain goes through rsqrtss in the reference build: 2 ulp across CPUs, exact IEEE variant within 4 ulp
    kb = k1.arr("pilot_b")
    assert abs(float(kb[0]) - float(t["peak_b"][0])) <= 2 * np.spacing(np.float32(t["peak_b"][0]))
    k_ieee = O.design(1_024_000, rsqrt_mode=0).arr("pilot_b")
    assert abs(float(k_ieee[0]) - float(t["peak_b"][0])) <= 4 * np.spacing(np.float32(t["peak_b"][0]))
    assert kb[1] == 0 and kb[2] == 0
    # analytic sanity: Hilbert taps antisymmetric with zeros at even offsets, FIR LPF taps symmetric
    h = k1.arr("b_hilbert")
    assert np.all(h[0::2] == 0) and np.allclose(h, -h[::-1])
    assert np.array_equal(k1.arr("b_lpr"), k1.arr("b_lpr")[::-1]) or np.allclose(k1.arr("b_lpr"), k1.arr("b_lpr")[::-1], atol=1e-7)


def test_primitives(golden):
    import ctypes as C
    g = golden("prims.npz")
    L = O.lib()
    xc = np.ascontiguousarray(g["x_c"]); xr = np.ascontiguousarray(g["x_r"])
    P = O._ptr

    def three(n_total):
        return [0, n_total // 4, n_total // 4 + n_total // 8, n_total]

    def decim_c(nn, m, k):
        b = np.zeros(nn, np.float32); L.fmo_design_fir_lpf(P(b), nn, k)
        hist = np.zeros((nn, 2), np.float32); n_out = xc.shape[0] // m
        y = np.zeros((n_out, 2), np.float32); cuts = three(n_out)
        for a, e in zip(cuts[:-1], cuts[1:]):
            L.fmo_decim_c32(P(hist), P(b), nn, m, P(xc[a * m:]), P(y[a:]), e - a)
        return y.reshape(-1)

    assert bits_equal(decim_c(64, 4, np.float32(0.2375)), g["poly_c_m4_n64_cf32"])
    assert bits_equal(decim_c(64, 8, np.float32(0.11875)), g["poly_c_m8_n64_cf32"])
    assert bits_equal(decim_c(128, 4, np.float32(15000.0 / 64000.0)), g["poly_c_m4_n128_cf32"])
    assert bits_equal(decim_c(128, 8, np.float32(2000.0 / 64000.0)), g["poly_c_m8_n128_cf32"])

    b = np.zeros(64, np.float32); L.fmo_design_fir_lpf(P(b), 64, np.float32(0.475))
    hist = np.zeros(64, np.float32); n_out = xr.size // 2; y = np.zeros(n_out, np.float32); cuts = three(n_out)
    for a, e in zip(cuts[:-1], cuts[1:]):
        L.fmo_decim_f32(P(hist), P(b), 64, 2, P(xr[a * 2:]), P(y[a:]), e - a)
    assert bits_equal(y, g["poly_r_m2_n64_f32"])

    hb = np.zeros(65, np.float32); L.fmo_design_hilbert(P(hb), 65)
    hist = np.zeros(65, np.float32); y = np.zeros((xr.size, 2), np.float32); cuts = three(xr.size)
    for a, e in zip(cuts[:-1], cuts[1:]):
        L.fmo_hilbert(P(hist), P(hb), P(xr[a:]), P(y[a:]), e - a)
    assert bits_equal(y.reshape(-1), g["hilbert_cf32"])

    pb = np.zeros(3, np.float32); pa = np.zeros(3, np.float32)
    L.fmo_design_iir_peak(P(pb), P(pa), np.float32(19000.0 / 64000.0), np.float32(0.9999), 1)
    xn = np.zeros((3, 2), np.float32); yn = np.zeros((3, 2), np.float32); y = np.zeros_like(xc)
    L.fmo_iir_c32(P(pb), P(pa), 3, P(xn), P(yn), P(xc), P(y), xc.shape[0])
    ok = bits_equal(y.reshape(-1), g["iir_peak_cf32"])
    assert ok or rms(y.reshape(-1) - g["iir_peak_cf32"]) <= 1e-6 * rms(g["iir_peak_cf32"])

    lb = np.zeros(2, np.float32); la = np.zeros(2, np.float32); L.fmo_design_iir_lpf(P(lb), P(la), np.float32(0.0497))
    xn = np.zeros(2, np.float32); yn = np.zeros(2, np.float32); y = np.zeros_like(xr)
    L.fmo_iir_f32(P(lb), P(la), 2, P(xn), P(yn), P(xr), P(y), xr.size)
    assert bits_equal(y, g["iir_lpf_f32"])

    gain = C.c_float(0.1); nb = xc.shape[0] // 3; y = np.zeros((nb * 3, 2), np.float32); gains = []
    for i in range(3):
        gains.append(L.fmo_agc(C.byref(gain), 0.5, 0.2, P(xc[i * nb:]), P(y[i * nb:]), nb))
    assert bits_equal(np.array(gains, np.float32), g["agc_gain_f32"])
    assert bits_equal(y.reshape(-1), g["agc_cf32"])

    prev = C.c_float(0.0); y = np.zeros(xc.shape[0], np.float32); h = xc.shape[0] // 2
    gain_fm = O.design(1_024_000).fm_gain
    L.fmo_discriminator(C.byref(prev), gain_fm, P(xc), P(y), h)
    L.fmo_discriminator(C.byref(prev), gain_fm, P(xc[h:]), P(y[h:]), xc.shape[0] - h)
    assert bits_equal(y, g["fm_discriminator_f32"])

    cheb = np.array([L.fmo_chebyshev_sine(float(v)) for v in g["grid"]], np.float32)
    assert bits_equal(cheb, g["chebyshev_f32"])
    assert np.max(np.abs(cheb - np.sin(2 * np.pi * g["grid"].astype(np.float64)))) < 1e-6

    dt = np.ascontiguousarray(g["dt"]); n = dt.size - 1
    for harm, off, name in ((2.0, 0.3217, "harmonic2_cf32"), (3.0, 0.0, "harmonic3_cf32")):
        y = np.zeros((n, 2), np.float32)
        L.fmo_harmonic_mix(P(dt), P(xc), P(y), n, harm, off)
        assert bits_equal(y.reshape(-1), g[name]), describe_diff(y.reshape(-1), g[name])


def test_wrong_block_size_is_dropped():
    d = O.Demod(16384, 1_024_000)
    assert d.process_cf32(np.zeros((100, 2), np.float32)) == -1  # reference: silent return, no output
    assert d.process_cf32(np.ones((16384, 2), np.float32)) == 0


def test_rate_parametrisation_consistency():
    """256 kSa/s input == the 1.024 MSa/s chain with stage 1 removed: feed the oracle's own fm_in."""
    cap = synth.to_cf32(synth.fm_capture(4 * 16384, seed=5)["iq"])
    full = O.run_chain(cap, 16384, 1_024_000, u8=False, streams=["fm_in", "audio", "rds_sym", "fm_out_iq"])
    fm_in = full["fm_in"].reshape(-1, 2)
    low = O.run_chain(fm_in, 4096, 256_000, u8=False, streams=["audio", "rds_sym", "fm_out_iq"])
    assert bits_equal(low["audio"], full["audio"])
    assert bits_equal(low["fm_out_iq"], full["fm_out_iq"])
    assert np.array_equal(low["rds_count"], full["rds_count"])
    # 2.048 MSa/s: stage-1 decimation 8; chain runs and produces the same number of audio frames per second
    cap2 = synth.to_cf32(synth.fm_capture(4 * 32768, fs=2_048_000.0, seed=5)["iq"])
    hi = O.run_chain(cap2, 32768, 2_048_000, u8=False, streams=["audio"])
    assert hi["audio"].size == full["audio"].size
    assert np.all(np.isfinite(hi["audio"]))
