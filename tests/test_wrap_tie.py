"""CPU test (no GPU): the tolerance mode's table for the discriminator's wrap at exactly half a turn on u8 captures
(fmd_design_wrap_tie; fmd_kernels.hip wrap_tie_u8) against glibc's atan2f and the reference's wrap (src/fm_demod/fm_demod.cpp:36-43):
for every u8 sample the sign of the wrapped phase difference to a sample in exactly the opposite direction — of any length."""
import ctypes as C

import numpy as np

import oraclelib as O


def _table():
    import fmradio_loader
    L = fmradio_loader.load().load_library()
    bits = np.zeros(2048, np.uint32)
    L.fmd_design_wrap_tie.argtypes = [C.c_void_p]
    assert L.fmd_design_wrap_tie(bits.ctypes.data_as(C.c_void_p)) == 0
    return bits


def _reference_wrap(y0, x0, y1, x1):
    d = (O.libm_atan2f(y1, x1) - O.libm_atan2f(y0, x0)).astype(np.float32)
    pi = np.float32(np.pi)
    return np.where(d >= pi, d - np.float32(2) * pi, np.where(d <= -pi, d + np.float32(2) * pi, d))


def test_wrap_tie_table_is_the_references_decision():
    bits = _table()
    xr, yr = np.meshgrid(np.arange(256), np.arange(256))
    xr = xr.ravel(); yr = yr.ravel()
    x = (xr - 127).astype(np.float32); y = (yr - 127).astype(np.float32)
    keep = (x != 0) | (y != 0)
    key = (yr << 8) | xr
    got = ((bits[key >> 5] >> (key & 31)) & 1).astype(bool)
    for k in (1, 2, 3):                      # the opposite sample k times as long (or 1 / k): the decision depends on the direction only
        ok = keep & (np.abs(k * x) <= 128) & (np.abs(k * y) <= 128)
        w = _reference_wrap(y[ok], x[ok], np.float32(0) - k * y[ok], np.float32(0) - k * x[ok])       # (0 - v: samples are (float)u8 - 127, never -0)
        assert np.all(np.abs(np.abs(w) - np.float32(np.pi)) < 1e-6)            # it IS a tie: +-pi to the last bits
        assert np.array_equal(got[ok], w > 0), k
    # both outcomes occur (the table is not a constant)
    assert 0.3 < got[keep].mean() < 0.7
    # a zero sample next to one on the negative real axis: deterministic, handled in the kernel without the table
    assert _reference_wrap(np.float32([0]), np.float32([0]), np.float32([0]), np.float32([-5]))[0] < 0
    assert _reference_wrap(np.float32([0]), np.float32([-5]), np.float32([0]), np.float32([0]))[0] > 0
