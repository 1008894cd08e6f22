"""Minimal RDS block/group synchroniser used as the known-answer checker in tests.

Same coding constants as the reference's decoder (reference src/rds_decoder/rds_constants.h:15-28,
crc10.cpp:9-25): 26-bit blocks, CRC-10 g(x)=x^10+x^8+x^7+x^5+x^4+x^3+1, offset words A,B,C,D.
Given the byte stream a Manchester decoder emits (MSB first), returns the list of groups
(A, B, C, D data words) whose four blocks all pass the CRC in sequence.
"""
from __future__ import annotations

import numpy as np

POLY = 0x5B9
OFFSETS = (0x0FC, 0x198, 0x168, 0x1B4)


def syndrome(block26: int) -> int:
    reg = block26
    for bit in range(25, 9, -1):
        if reg & (1 << bit):
            reg ^= POLY << (bit - 10)
    return reg & 0x3FF


def decode_groups(rds_bytes: np.ndarray) -> list[tuple[int, int, int, int]]:
    bits = np.unpackbits(np.asarray(rds_bytes, dtype=np.uint8))
    n = bits.size
    groups = []
    i = 0
    while i + 104 <= n:
        words = []
        ok = True
        for blk in range(4):
            v = 0
            for b in bits[i + 26 * blk: i + 26 * (blk + 1)]:
                v = (v << 1) | int(b)
            if syndrome(v ^ OFFSETS[blk]) != 0:
                ok = False
                break
            words.append(v >> 10)
        if ok:
            groups.append(tuple(words))
            i += 104
        else:
            i += 1
    return groups
