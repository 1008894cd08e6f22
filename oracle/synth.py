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
