"""Synthetic broadcast-FM capture generator (TEST INFRASTRUCTURE, not product code).

The reference ships no recordings (its `sample_data/*.pcm` are a release asset,
reference README.md:56-60) and no tests, so parity runs use captures synthesised
here: stereo tones + 19 kHz pilot + L-R on 38 kHz DSB-SC + valid RDS 0A groups on
57 kHz BPSK, frequency-modulated with 75 kHz deviation, optionally quantised to
the RTL-SDR u8 IQ format the reference ingests (reference src/rtl_sdr.cpp:42,
src/app.cpp:56-62: `(float)u8 - 127`).

RDS coding follows the constants the reference's decoder uses
(reference src/rds_decoder/rds_constants.h:15-28, crc10.cpp:9-25): 26-bit blocks
= 16 data bits + CRC-10 (g(x)=x^10+x^8+x^7+x^5+x^4+x^3+1) xor offset word
A/B/C/D, MSB first, differentially encoded, biphase symbols at 2375 sym/s.

Everything is float64 numpy with a seeded PCG64 generator.
"""
from __future__ import annotations

import numpy as np

RDS_OFFSET = {"A": 0x0FC, "B": 0x198, "C": 0x168, "D": 0x1B4}
RDS_POLY = 0x5B9  # x^10 + x^8 + x^7 + x^5 + x^4 + x^3 + 1
RDS_SYMBOL_RATE = 2375.0  # biphase symbols / s (bit rate 1187.5)


def rds_crc10(data16: int) -> int:
    """Remainder of data16 * x^10 modulo g(x)."""
    reg = data16 << 10
    for bit in range(25, 9, -1):
        if reg & (1 << bit):
            reg ^= RDS_POLY << (bit - 10)
    return reg & 0x3FF


def rds_block(data16: int, offset: str) -> list[int]:
    word = (data16 << 10) | (rds_crc10(data16) ^ RDS_OFFSET[offset])
    return [(word >> (25 - i)) & 1 for i in range(26)]


def rds_group_0a(pi: int, segment: int, ps_name: str = "MI355XFM", af: int = 0xE0CD) -> tuple[list[int], tuple[int, int, int, int]]:
    """One type-0A group (PI, flags+segment, AF pair, two PS characters)."""
    seg = segment & 3
    blk_b = (0x0 << 12) | (0 << 11) | (0 << 10) | (0 << 5) | (0 << 4) | (1 << 3) | (0 << 2) | seg
    chars = ps_name.encode("ascii")[2 * seg : 2 * seg + 2]
    blk_d = (chars[0] << 8) | chars[1]
    words = (pi & 0xFFFF, blk_b, af, blk_d)
    bits = rds_block(words[0], "A") + rds_block(words[1], "B") + rds_block(words[2], "C") + rds_block(words[3], "D")
    return bits, words


def rds_bitstream(n_bits: int, pi: int) -> tuple[np.ndarray, list[tuple[int, int, int, int]]]:
    bits: list[int] = []
    groups = []
    seg = 0
    while len(bits) < n_bits:
        b, w = rds_group_0a(pi, seg)
        bits += b
        groups.append(w)
        seg += 1
    return np.array(bits[:n_bits], dtype=np.uint8), groups


def fm_capture(
    n_samples: int,
    fs: float = 1_024_000.0,
    seed: int = 1234,
    channel: int = 0,
    noise_sigma: float = 0.02,
    jitter: bool = True,
    pilot_hz: float = 19000.0,
    pilot_level: float = 0.10,
) -> dict:
    """Complex baseband FM capture of one station.

    Returns dict(iq=complex128[n], groups=[(A,B,C,D)...], pi=int).
    Channel `c` uses seed+c, PI 0x1234+c and (if jitter) tone frequencies
    shifted by a few percent so batched channels are not identical.
    """
    rng = np.random.default_rng(seed + channel)
    t = np.arange(n_samples, dtype=np.float64) / fs
    if jitter and channel != 0:
        j = 1.0 + 0.05 * rng.uniform(-1.0, 1.0, size=4)
        ph = rng.uniform(0.0, 2.0 * np.pi, size=4)
    else:
        j = np.ones(4)
        ph = np.zeros(4)
    two_pi = 2.0 * np.pi
    left = 0.5 * np.sin(two_pi * 1000.0 * j[0] * t + ph[0]) + 0.3 * np.sin(two_pi * 3300.0 * j[1] * t + ph[1])
    right = 0.5 * np.sin(two_pi * 440.0 * j[2] * t + ph[2]) + 0.3 * np.sin(two_pi * 5000.0 * j[3] * t + ph[3])

    pi_code = (0x1234 + channel) & 0xFFFF
    n_sym = int(np.ceil(n_samples / fs * RDS_SYMBOL_RATE)) + 4
    n_bits = n_sym // 2 + 2
    bits, groups = rds_bitstream(n_bits, pi_code)
    diff = np.bitwise_xor.accumulate(bits)  # differential encoding
    lvl = 2.0 * diff.astype(np.float64) - 1.0
    sym = np.empty(2 * n_bits, dtype=np.float64)  # biphase: d -> (+d, -d)
    sym[0::2] = lvl
    sym[1::2] = -lvl
    sym_idx = np.floor(t * RDS_SYMBOL_RATE).astype(np.int64)
    rds = sym[sym_idx]

    p = two_pi * pilot_hz * t   # (pilot_hz / pilot_level other than the defaults: stress cases for the PLL tests)
    mpx = (
        0.40 * (left + right) / 1.6
        + pilot_level * np.sin(p)
        + 0.40 * (left - right) / 1.6 * np.sin(2.0 * p)
        + 0.06 * rds * np.sin(3.0 * p)
    )
    phase = two_pi * 75000.0 * np.cumsum(mpx) / fs
    iq = np.exp(1j * phase)
    if noise_sigma > 0.0:
        iq = iq + noise_sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    return {"iq": iq, "groups": groups, "pi": pi_code}


def to_u8(iq: np.ndarray) -> np.ndarray:
    """RTL-SDR style interleaved u8 IQ: round(127 + 100*x), clipped. Shape [n, 2]."""
    out = np.empty((iq.shape[0], 2), dtype=np.float64)
    out[:, 0] = iq.real
    out[:, 1] = iq.imag
    return np.clip(np.rint(127.0 + 100.0 * out), 0, 255).astype(np.uint8)


def u8_to_cf32(u8: np.ndarray) -> np.ndarray:
    """The reference's ingest conversion (src/app.cpp:56-62): float(u8) - 127, no scaling."""
    f = u8.astype(np.float32) - np.float32(127.0)
    return f.reshape(-1, 2)


def to_cf32(iq: np.ndarray, scale: float = 100.0) -> np.ndarray:
    """Float capture scaled like the u8 path (x100) without quantisation. Shape [n, 2] float32."""
    out = np.empty((iq.shape[0], 2), dtype=np.float32)
    out[:, 0] = (scale * iq.real).astype(np.float32)
    out[:, 1] = (scale * iq.imag).astype(np.float32)
    return out


# ---- realistic programme and channel impairments (round 5) ---------------------------------------------------
# The reference's recordings are a release asset (reference README.md:56-60) and not in the tree, so the synthesiser supplies what a
# real capture has and fm_capture() lacks: a noise-like, pre-emphasised programme up to 15 kHz with independent L and R, a carrier
# that is not at 0 Hz, a range of carrier-to-noise ratios, slow amplitude fading, an adjacent station, over-deviation, and RDS
# traffic of more than one group type.  Float64 numpy, seeded like fm_capture().

def rds_group_2a(pi: int, segment: int, text: str = "ROUND FIVE REALISTIC PROGRAMME TEXT ON MI355X, RDS GROUP 2A.     ") -> tuple[list[int], tuple[int, int, int, int]]:
    """One type-2A group (radiotext segment of four characters; reference decoder: src/rds_decoder/rds_decoder.cpp group 2 handler)."""
    seg = segment & 15
    blk_b = (0x2 << 12) | (0 << 11) | (0 << 10) | (0 << 5) | (0 << 4) | seg
    chars = text.encode("ascii")[4 * seg: 4 * seg + 4]
    words = (pi & 0xFFFF, blk_b, (chars[0] << 8) | chars[1], (chars[2] << 8) | chars[3])
    bits = rds_block(words[0], "A") + rds_block(words[1], "B") + rds_block(words[2], "C") + rds_block(words[3], "D")
    return bits, words


def rds_group_4a(pi: int, minute: int) -> tuple[list[int], tuple[int, int, int, int]]:
    """One type-4A group (clock time and date: MJD 60586, a UTC time that advances with `minute`)."""
    mjd, hour, mnt = 60586, (12 + minute // 60) % 24, minute % 60
    blk_b = (0x4 << 12) | (0 << 11) | (0 << 10) | (0 << 5) | ((mjd >> 15) & 3)
    blk_c = ((mjd & 0x7FFF) << 1) | ((hour >> 4) & 1)
    blk_d = ((hour & 15) << 12) | (mnt << 6) | 0
    words = (pi & 0xFFFF, blk_b, blk_c & 0xFFFF, blk_d & 0xFFFF)
    bits = rds_block(words[0], "A") + rds_block(words[1], "B") + rds_block(words[2], "C") + rds_block(words[3], "D")
    return bits, words


def rds_bitstream_mixed(n_bits: int, pi: int) -> tuple[np.ndarray, list[tuple[int, int, int, int]]]:
    """0A, 2A, 0A, 2A, ..., a 4A every 16 groups: the mix a station with PS name, radiotext and clock sends."""
    bits: list[int] = []
    groups = []
    k = 0
    while len(bits) < n_bits:
        if k % 16 == 15:
            b, w = rds_group_4a(pi, k // 16)
        elif k & 1:
            b, w = rds_group_2a(pi, k // 2)
        else:
            b, w = rds_group_0a(pi, k // 2)
        bits += b
        groups.append(w)
        k += 1
    return np.array(bits[:n_bits], dtype=np.uint8), groups


def _band_noise(rng, n: int, fs: float, lo_hz: float, hi_hz: float, tau_us: float) -> np.ndarray:
    """Gaussian noise between lo_hz and hi_hz (raised-cosine edges 500 Hz wide) through the pre-emphasis 1 + j 2 pi f tau."""
    spec = np.fft.rfft(rng.standard_normal(n))
    f = np.fft.rfftfreq(n, 1.0 / fs)
    edge = 500.0
    w = np.clip((f - (lo_hz - edge / 2)) / edge, 0.0, 1.0) * np.clip(((hi_hz + edge / 2) - f) / edge, 0.0, 1.0)
    w = 0.5 - 0.5 * np.cos(np.pi * w)
    # a programme's long-term spectrum falls ~6 dB / octave above ~2 kHz; the pre-emphasis lifts it back
    w = w / np.sqrt(1.0 + (f / 2000.0) ** 2)
    if tau_us > 0.0:
        w = w * np.abs(1.0 + 2j * np.pi * f * tau_us * 1e-6)
    return np.fft.irfft(spec * w, n)


def programme(rng, n: int, fs: float, tau_us: float = 75.0, peak: float = 0.8, correlation: float = 0.3) -> tuple[np.ndarray, np.ndarray]:
    """Left and right audio of a noise-like programme to 15 kHz: a common part (`correlation` of the power) and independent parts,
    pre-emphasised, level set so that 0.1 % of the samples would exceed `peak`, then limited to it (what a broadcast limiter does)."""
    common = _band_noise(rng, n, fs, 40.0, 15000.0, tau_us)
    out = []
    for _ in range(2):
        own = _band_noise(rng, n, fs, 40.0, 15000.0, tau_us)
        x = np.sqrt(correlation) * common + np.sqrt(1.0 - correlation) * own
        x = x * (peak / np.quantile(np.abs(x), 0.999))
        out.append(np.clip(x, -peak, peak))
    return out[0], out[1]


def fm_capture_realistic(
    n_samples: int,
    fs: float = 256_000.0,
    seed: int = 1234,
    channel: int = 0,
    cnr_db: float = 40.0,
    carrier_offset_hz: float = 0.0,
    deviation_hz: float = 75000.0,
    preemphasis_us: float = 75.0,
    fading_hz: float = 0.0,
    fading_k_db: float = 6.0,
    adjacent_db: float | None = None,
    adjacent_offset_hz: float = 200_000.0,
    pilot_hz: float = 19000.0,
    mixed_groups: bool = True,
) -> dict:
    """Complex baseband capture of one station with a realistic programme and channel (see the section comment).

    MPX as fm_capture(): 0.40 (L+R)/1.6 + 0.10 pilot + 0.40 (L-R)/1.6 on 38 kHz + 0.06 RDS on 57 kHz, times deviation_hz (over-deviation:
    > 75 kHz).  cnr_db: carrier power over the noise power in the capture's bandwidth fs.  fading_hz > 0: Rician amplitude fading
    (K = fading_k_db) with that Doppler spread.  adjacent_db: a second station, that many dB BELOW this one (negative = stronger),
    adjacent_offset_hz away (needs fs / 2 > offset + its deviation: 1.024 MSa/s captures).
    Returns dict(iq=complex128[n], groups=[...], pi=int)."""
    rng = np.random.default_rng(seed + 7919 * channel)
    t = np.arange(n_samples, dtype=np.float64) / fs
    two_pi = 2.0 * np.pi
    left, right = programme(rng, n_samples, fs, preemphasis_us)
    pi_code = (0x1234 + channel) & 0xFFFF
    n_sym = int(np.ceil(n_samples / fs * RDS_SYMBOL_RATE)) + 4
    n_bits = n_sym // 2 + 2
    bits, groups = (rds_bitstream_mixed if mixed_groups else rds_bitstream)(n_bits, pi_code)
    diff = np.bitwise_xor.accumulate(bits)
    lvl = 2.0 * diff.astype(np.float64) - 1.0
    sym = np.empty(2 * n_bits, dtype=np.float64)
    sym[0::2] = lvl
    sym[1::2] = -lvl
    rds = sym[np.floor(t * RDS_SYMBOL_RATE).astype(np.int64)]
    ph0 = rng.uniform(0.0, two_pi)
    p = two_pi * pilot_hz * t + ph0
    mpx = 0.40 * (left + right) / 1.6 + 0.10 * np.sin(p) + 0.40 * (left - right) / 1.6 * np.sin(2.0 * p) + 0.06 * rds * np.sin(3.0 * p)
    phase = two_pi * deviation_hz * np.cumsum(mpx) / fs + two_pi * carrier_offset_hz * t
    iq = np.exp(1j * phase)
    if fading_hz > 0.0:
        spec = np.fft.fft(rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
        f = np.fft.fftfreq(n_samples, 1.0 / fs)
        g = np.fft.ifft(spec * (np.abs(f) <= fading_hz))
        g = g / np.sqrt(np.mean(np.abs(g) ** 2))
        kf = 10.0 ** (fading_k_db / 10.0)
        iq = iq * ((np.sqrt(kf) + g) / np.sqrt(kf + 1.0))
    if adjacent_db is not None:
        if fs / 2.0 < abs(adjacent_offset_hz) + 100e3:
            raise ValueError("the adjacent station does not fit the capture's bandwidth")
        l2, r2 = programme(rng, n_samples, fs, preemphasis_us)
        p2 = two_pi * 19000.0 * t + rng.uniform(0.0, two_pi)
        mpx2 = 0.45 * (l2 + r2) / 1.6 + 0.10 * np.sin(p2) + 0.45 * (l2 - r2) / 1.6 * np.sin(2.0 * p2)
        iq = iq + 10.0 ** (-adjacent_db / 20.0) * np.exp(1j * (two_pi * 75000.0 * np.cumsum(mpx2) / fs + two_pi * adjacent_offset_hz * t))
    sigma = np.sqrt(0.5 * 10.0 ** (-cnr_db / 10.0))
    iq = iq + sigma * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    return {"iq": iq, "groups": groups, "pi": pi_code}


# the conditions tests/test_gpu_realistic.py runs (name -> keyword arguments of fm_capture_realistic)
REALISTIC_CONDITIONS = {
    "programme_cnr40": dict(cnr_db=40.0),
    "programme_cnr25": dict(cnr_db=25.0),
    "programme_cnr15": dict(cnr_db=15.0),
    "preemphasis_50us": dict(cnr_db=35.0, preemphasis_us=50.0),
    "carrier_plus_30k": dict(cnr_db=35.0, carrier_offset_hz=30000.0),
    "carrier_minus_30k": dict(cnr_db=35.0, carrier_offset_hz=-30000.0),
    "overdeviation_110k": dict(cnr_db=35.0, deviation_hz=110000.0),
    "fading_5hz": dict(cnr_db=30.0, fading_hz=5.0),
    "pilot_plus_2hz": dict(cnr_db=35.0, pilot_hz=19002.0),
    "adjacent_minus_20db": dict(cnr_db=35.0, adjacent_db=20.0),       # 1.024 MSa/s only
}
