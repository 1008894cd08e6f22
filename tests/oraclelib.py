"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE: the CPU restatement of the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LIB_PATH = ORACLE_DIR / "liboracle.so"
REF_DUMP = ORACLE_DIR / "_ref" / "fm_ref_dump"
REF_BENCH = ORACLE_DIR / "_ref" / "fm_demod_benchmark"


class Controls(C.Structure):
    _fields_ = [
        ("audio_out", C.c_int),
        ("audio_stereo_mix_factor", C.c_float),
        ("use_deemphasis", C.c_int),
        ("deemphasis_tus", C.c_int),
        ("lpr_cutoff_hz", C.c_int),
        ("lmr_cutoff_hz", C.c_int),
    ]


class Coeffs(C.Structure):
    _fields_ = [
        ("fs_baseband", C.c_int),
        ("m_fm_in", C.c_int),
        ("b_fm_in", C.c_float * 64),
        ("b_fm_out", C.c_float * 64),
        ("b_hilbert", C.c_float * 65),
        ("pilot_b", C.c_float * 3),
        ("pilot_a", C.c_float * 3),
        ("pll_lpf_b", C.c_float * 2),
        ("pll_lpf_a", C.c_float * 2),
        ("deemph_b", C.c_float * 2),
        ("deemph_a", C.c_float * 2),
        ("b_lpr", C.c_float * 128),
        ("b_lmr", C.c_float * 128),
        ("b_rds", C.c_float * 128),
        ("ted_lpf_b", C.c_float * 2),
        ("ted_lpf_a", C.c_float * 2),
        ("bpsk_lpf_b", C.c_float * 2),
        ("bpsk_lpf_a", C.c_float * 2),
        ("fm_gain", C.c_float),
    ]

    def arr(self, name: str) -> np.ndarray:
        v = getattr(self, name)
        return np.ctypeslib.as_array(v).copy() if hasattr(v, "__len__") else np.float32(v)


class Manchester(C.Structure):
    _fields_ = [("buf", C.c_uint8 * 16), ("byte_index", C.c_int), ("bit_index", C.c_int), ("is_read_bit", C.c_int), ("prev_bit", C.c_int)]


class CF32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


def build() -> None:
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR), "oracle"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src_newer = (not LIB_PATH.exists()) or any(
        (ORACLE_DIR / f).stat().st_mtime > LIB_PATH.stat().st_mtime for f in ("fm_oracle.c", "fm_oracle.h")
    )
    if src_newer:
        build()
    L = C.CDLL(str(LIB_PATH))
    fp = C.POINTER(C.c_float)
    L.fmo_default_controls.argtypes = [C.POINTER(Controls)]
    L.fmo_design.argtypes = [C.POINTER(Coeffs), C.c_int, C.POINTER(Controls), C.c_int]
    L.fmo_create.restype = C.c_void_p
    L.fmo_create.argtypes = [C.c_int, C.c_int]
    L.fmo_destroy.argtypes = [C.c_void_p]
    L.fmo_set_controls.argtypes = [C.c_void_p, C.POINTER(Controls)]
    L.fmo_set_coeffs.argtypes = [C.c_void_p, C.POINTER(Coeffs)]
    L.fmo_get_coeffs.argtypes = [C.c_void_p, C.POINTER(Coeffs)]
    L.fmo_process_cf32.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_process_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_get.restype = fp
    L.fmo_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
    L.fmo_rds_symbol_count.argtypes = [C.c_void_p]
    L.fmo_manchester_init.argtypes = [C.POINTER(Manchester)]
    L.fmo_manchester_push.argtypes = [C.POINTER(Manchester), C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.fmo_chebyshev_sine.restype = C.c_float
    L.fmo_chebyshev_sine.argtypes = [C.c_float]
    L.fmo_atan2f_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    L.fmo_dot_f32.restype = C.c_float
    L.fmo_dot_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_dot_c32.restype = CF32
    L.fmo_dot_c32.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_decim_c32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_decim_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_hilbert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_iir_c32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_iir_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_agc.restype = C.c_float
    L.fmo_agc.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_discriminator.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_void_p, C.c_void_p, C.c_int]
    L.fmo_harmonic_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
    L.fmo_design_fir_lpf.argtypes = [C.c_void_p, C.c_int, C.c_float]
    L.fmo_design_hilbert.argtypes = [C.c_void_p, C.c_int]
    L.fmo_design_iir_lpf.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    L.fmo_design_iir_peak.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int]
    _lib = L
    return L


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def libm_atan2f(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Host libm atan2f (glibc), elementwise."""
    y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(y)
    lib().fmo_atan2f_array(_ptr(y), _ptr(x), _ptr(out), y.size)
    return out


def default_controls() -> Controls:
    c = Controls()
    lib().fmo_default_controls(C.byref(c))
    return c


def design(fs_baseband: int = 1_024_000, controls: Controls | None = None, rsqrt_mode: int = 1) -> Coeffs:
    k = Coeffs()
    c = controls if controls is not None else default_controls()
    lib().fmo_design(C.byref(k), fs_baseband, C.byref(c), rsqrt_mode)
    return k


STREAMS = [
    "fm_in", "fm_demod", "fm_out", "fm_out_iq", "pilot", "pll_dt", "pll", "pll_raw_err", "pll_pi_err",
    "lpr", "lmr", "rds", "rds_raw_sym", "rds_sym", "audio", "lmr_phase",
    "bpsk_pll_sym", "bpsk_intdump", "bpsk_ted_raw", "bpsk_ted_pi", "bpsk_pll_raw", "bpsk_pll_pi", "bpsk_zcd", "bpsk_trig",
]


class Demod:
    """One-channel oracle demodulator; mirrors Broadcast_FM_Demod (reference broadcast_fm_demod.h:229-298)."""

    def __init__(self, block_size: int = 65536, fs_baseband: int = 1_024_000):
        self.L = lib()
        self.h = self.L.fmo_create(block_size, fs_baseband)
        if not self.h:
            raise ValueError("fmo_create failed")
        self.block_size = block_size
        self.man = Manchester()
        self.L.fmo_manchester_init(C.byref(self.man))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.fmo_destroy(self.h)
            self.h = None

    def set_controls(self, c: Controls):
        self.L.fmo_set_controls(self.h, C.byref(c))

    def set_coeffs(self, k: Coeffs):
        self.L.fmo_set_coeffs(self.h, C.byref(k))

    def get_coeffs(self) -> Coeffs:
        k = Coeffs()
        self.L.fmo_get_coeffs(self.h, C.byref(k))
        return k

    def process_cf32(self, iq: np.ndarray) -> int:
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        return self.L.fmo_process_cf32(self.h, _ptr(iq), iq.size // 2)

    def process_u8(self, iq: np.ndarray) -> int:
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        return self.L.fmo_process_u8(self.h, _ptr(iq), iq.size // 2)

    def get(self, name: str) -> np.ndarray:
        n = C.c_int(0)
        p = self.L.fmo_get(self.h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def manchester(self, sym: np.ndarray) -> bytes:
        sym = np.ascontiguousarray(sym, dtype=np.float32)
        out = np.zeros(((sym.size // 256) + 2) * 16, dtype=np.uint8)
        n = self.L.fmo_manchester_push(C.byref(self.man), _ptr(sym), sym.size, _ptr(out), out.size)
        return out[:n].tobytes()


def run_chain(capture: np.ndarray, block_size: int = 65536, fs_baseband: int = 1_024_000, u8: bool = True,
              controls: Controls | None = None, coeffs: Coeffs | None = None, streams=None) -> dict:
    """Run whole blocks of `capture` ([n,2] u8 or float32) through the oracle; concatenated per-block dumps."""
    d = Demod(block_size, fs_baseband)
    if controls is not None:
        d.set_controls(controls)
    if coeffs is not None:
        d.set_coeffs(coeffs)
    names = list(streams) if streams is not None else STREAMS
    out = {k: [] for k in names}
    out["rds_count"] = []
    rds_bytes = b""
    nb = capture.shape[0] // block_size
    for b in range(nb):
        blk = capture[b * block_size:(b + 1) * block_size]
        rc = d.process_u8(blk) if u8 else d.process_cf32(blk)
        assert rc == 0
        for k in names:
            out[k].append(d.get(k))
        out["rds_count"].append(d.L.fmo_rds_symbol_count(d.h))
        rds_bytes += d.manchester(d.get("rds_sym"))
    res = {k: np.concatenate(v) if len(v) else np.zeros(0, np.float32) for k, v in out.items() if k != "rds_count"}
    res["rds_count"] = np.array(out["rds_count"], dtype=np.int32)
    res["rds_bytes"] = np.frombuffer(rds_bytes, dtype=np.uint8)
    res["coeffs"] = d.get_coeffs()
    return res


# ---- the real reference, compiled (oracle/_ref) -------------------------------------------------

REF_FILES = {
    "fm_out_iq": ("fm_out_iq.cf32", np.float32), "pilot": ("pilot.cf32", np.float32), "pll": ("pll.cf32", np.float32),
    "pll_raw_err": ("pll_raw_err.f32", np.float32), "pll_pi_err": ("pll_pi_err.f32", np.float32),
    "lpr": ("lpr.f32", np.float32), "lmr": ("lmr.f32", np.float32), "rds": ("rds.cf32", np.float32),
    "rds_raw_sym": ("rds_raw_sym.cf32", np.float32), "rds_sym": ("rds_sym.f32", np.float32),
    "rds_count": ("rds_count.i32", np.int32), "audio": ("audio.f32", np.float32), "lmr_phase": ("lmr_phase.f32", np.float32),
    "rds_bytes": ("rds_bytes.u8", np.uint8),
    "bpsk_pll_sym": ("bpsk_pll_sym.cf32", np.float32), "bpsk_intdump": ("bpsk_intdump.cf32", np.float32),
    "bpsk_ted_raw": ("bpsk_ted_raw.f32", np.float32), "bpsk_ted_pi": ("bpsk_ted_pi.f32", np.float32),
    "bpsk_pll_raw": ("bpsk_pll_raw.f32", np.float32), "bpsk_pll_pi": ("bpsk_pll_pi.f32", np.float32),
    "bpsk_zcd": ("bpsk_zcd.u8", np.uint8), "bpsk_trig": ("bpsk_trig.u8", np.uint8),
}


def have_ref() -> bool:
    return REF_DUMP.exists() and os.access(REF_DUMP, os.X_OK)


def run_ref_chain(capture: np.ndarray, tmpdir, block_size: int = 65536, u8: bool = True, extra_args=()) -> dict:
    """Run the compiled reference (oracle/_ref/fm_ref_dump) on a capture and load its dumps."""
    tmpdir = Path(tmpdir)
    tmpdir.mkdir(parents=True, exist_ok=True)
    cap = tmpdir / ("cap.u8" if u8 else "cap.cf32")
    np.ascontiguousarray(capture, dtype=np.uint8 if u8 else np.float32).tofile(cap)
    out = tmpdir / "out"
    out.mkdir(exist_ok=True)
    mode = "chain" if u8 else "cf32chain"
    subprocess.run([str(REF_DUMP), mode, str(cap), str(out), str(block_size), *map(str, extra_args)],
                   check=True, stderr=subprocess.DEVNULL)
    return {k: np.fromfile(out / f, dtype=dt) for k, (f, dt) in REF_FILES.items()}
