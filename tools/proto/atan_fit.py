#!/usr/bin/env python3
"""Coefficients of the tolerance mode's arctangents (fm-radio_amd/csrc/fmd_math.h): atan(a) = a P(a^2) on [0, 1], P of n coefficients,
fitted by reweighted least squares on Chebyshev nodes (converges to the minimax polynomial), evaluated as the kernels do in float32."""
import sys
import numpy as np


def fit(nc: int, iters: int = 60):
    a = np.cos(np.linspace(0, np.pi, 4001)) * 0.5 + 0.5
    a = a[a > 1e-6]
    z = a * a
    A = np.stack([a * z ** k for k in range(nc)], 1)
    y = np.arctan(a)
    w = np.ones_like(a)
    for _ in range(iters):
        c = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]
        e = np.abs(A @ c - y)
        w *= 1 + 2 * e / e.max()
        w /= w.mean()
    return c


for nc in [int(v) for v in sys.argv[1:]] or [6, 8]:
    c32 = fit(nc).astype(np.float32)
    a = np.linspace(0, 1, 200001, dtype=np.float32)
    z = a * a
    p = np.zeros_like(a)
    for k in range(nc - 1, -1, -1):
        p = p * z + c32[k]
    err = np.abs((p * a).astype(np.float64) - np.arctan(a.astype(np.float64))).max()
    print(f"{nc} coefficients: max error {err:.3e} rad = {err / 2 / np.pi:.3e} turns;", ", ".join(repr(float(v)) for v in c32))
