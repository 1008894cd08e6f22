"""ctypes binding of libfmdemod.so (include/fmdemod.h) + a small batched-demodulator wrapper.

`BatchDemod` mirrors the reference's `Broadcast_FM_Demod` usage (construct with a block size, call
`process`, read audio / RDS symbols; reference src/fm_demod/broadcast_fm_demod.h:229-298) for C
channels at once.  It only moves pointers: inputs may be torch CUDA tensors (zero-copy, device entry
points) or numpy arrays (host entry points).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
HEADER = ROOT / "include" / "fmdemod.h"
DEBUG_HEADER = ROOT / "include" / "fmdemod_debug.h"   # self-test / profiling hooks: not part of the drop-in boundary

FMD_AUDIO_LPR, FMD_AUDIO_LMR, FMD_AUDIO_STEREO = 0, 1, 2
FMD_FLAG_KEEP_TAPS = 1
FMD_FLAG_NO_PIPELINE = 2
FMD_FLAG_PLL_TIME_PARALLEL = 4
FMD_FLAG_PLL_LOW_WORK = 8
FMD_FLAG_PLL_K8 = 16
FMD_FLAG_PLL_STREAM_ORDER = 32
FMD_FLAG_FAST_MATH = 64
FMD_OK, FMD_ERR_ARG, FMD_ERR_SIZE, FMD_ERR_DEVICE, FMD_ERR_NO_DEVICE, FMD_ERR_NAME, FMD_ERR_STATE = 0, -1, -2, -3, -4, -5, -6
FMD_OUTPUT_LIFETIME_BLOCKS = 5   # include/fmdemod.h; checked against the loaded library in load_library()


class FmdError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"fmdemod status {status}: {msg}")
        self.status = status


class Config(C.Structure):
    _fields_ = [("n_channels", C.c_int), ("block_size", C.c_int), ("fs_baseband", C.c_int), ("device", C.c_int), ("flags", C.c_uint)]


class Controls(C.Structure):
    _fields_ = [("audio_out", C.c_int), ("audio_stereo_mix_factor", C.c_float), ("use_deemphasis", C.c_int),
                ("deemphasis_tus", C.c_int), ("lpr_cutoff_hz", C.c_int), ("lmr_cutoff_hz", C.c_int)]


class Rates(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("fs_baseband", "fs_fm_in", "fs_fm_out", "fs_rds", "fs_audio",
                                       "n_baseband", "n_fm_in", "n_fm_out", "n_rds", "n_audio")]


class Coeffs(C.Structure):
    _fields_ = [("fs_baseband", C.c_int), ("m_fm_in", C.c_int), ("b_fm_in", C.c_float * 64), ("b_fm_out", C.c_float * 64),
                ("b_hilbert", C.c_float * 65), ("pilot_b", C.c_float * 3), ("pilot_a", C.c_float * 3),
                ("pll_lpf_b", C.c_float * 2), ("pll_lpf_a", C.c_float * 2), ("deemph_b", C.c_float * 2), ("deemph_a", C.c_float * 2),
                ("b_lpr", C.c_float * 128), ("b_lmr", C.c_float * 128), ("b_rds", C.c_float * 128),
                ("ted_lpf_b", C.c_float * 2), ("ted_lpf_a", C.c_float * 2), ("bpsk_lpf_b", C.c_float * 2), ("bpsk_lpf_a", C.c_float * 2),
                ("fm_gain", C.c_float)]


class ChanConfig(C.Structure):
    _fields_ = [("fs_in", C.c_double), ("fs_out", C.c_double), ("n_stations", C.c_int), ("center_hz", C.POINTER(C.c_double)),
                ("taps_per_phase", C.c_int), ("max_input_samples", C.c_longlong), ("device", C.c_int)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("total_ms", C.c_double), ("launches", C.c_int)]


def lib_path() -> Path:
    return CSRC / "libfmdemod.so"


def build_library(force: bool = False) -> Path:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-s", "-C", str(CSRC), "clean"], check=True)
    subprocess.run(["make", "-s", "-C", str(CSRC)], check=True)
    return lib_path()


def declared_symbols(debug: bool = True) -> list[str]:
    """Every function include/fmdemod.h declares (+ include/fmdemod_debug.h's hooks)."""
    names = set()
    for hdr in (HEADER, DEBUG_HEADER) if debug else (HEADER,):
        text = re.sub(r"/\*.*?\*/", "", hdr.read_text(), flags=re.S)
        names |= set(re.findall(r"\b(fmd_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


_lib = None


def load_library():
    """Load libfmdemod.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise FileNotFoundError(f"{p} missing: run __graft_entry__.build() / make -C {CSRC}")
    # the pipeline uses 5 streams beside the caller's: give them their own hardware queues (default is 4 per process)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    try:
        # torch bundles its own HIP runtime; load it FIRST so this process ends up with a single libamdhip64
        # (two runtimes in one process do not see each other's devices, streams or allocations)
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(str(p))
    H = C.c_void_p
    L.fmd_api_version.restype = C.c_int
    L.fmd_status_string.restype = C.c_char_p
    L.fmd_status_string.argtypes = [C.c_int]
    L.fmd_device_count.restype = C.c_int
    L.fmd_default_controls.argtypes = [C.POINTER(Controls)]
    L.fmd_default_config.argtypes = [C.POINTER(Config), C.c_int, C.c_int]
    L.fmd_create.argtypes = [C.POINTER(Config), C.POINTER(H)]
    L.fmd_destroy.argtypes = [H]
    L.fmd_reset.argtypes = [H]
    L.fmd_set_controls.argtypes = [H, C.c_int, C.POINTER(Controls)]
    L.fmd_get_controls.argtypes = [H, C.c_int, C.POINTER(Controls)]
    L.fmd_get_rates.argtypes = [H, C.POINTER(Rates)]
    L.fmd_get_coeffs.argtypes = [H, C.c_int, C.POINTER(Coeffs)]
    for name in ("fmd_process_cf32_dev", "fmd_process_u8_dev", "fmd_submit_cf32_dev", "fmd_submit_u8_dev"):
        getattr(L, name).argtypes = [H, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for name in ("fmd_process_cf32_host", "fmd_process_u8_host"):
        getattr(L, name).argtypes = [H, C.c_void_p, C.c_int, C.c_int]
    L.fmd_synchronize.argtypes = [H]
    L.fmd_wait_outputs.argtypes = [H, C.c_void_p]
    L.fmd_set_output_lag.argtypes = [H, C.c_int]
    L.fmd_outputs_block.argtypes = [H, C.POINTER(C.c_long)]
    L.fmd_wait_input.argtypes = [H, C.c_void_p]
    L.fmd_release_outputs.argtypes = [H, C.c_void_p]
    L.fmd_output_lifetime_blocks.restype = C.c_int
    L.fmd_state_size.restype = C.c_size_t
    L.fmd_state_size.argtypes = [H]
    L.fmd_get_state.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t]
    L.fmd_set_state.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t]
    L.fmd_audio_dev.argtypes = [H, C.POINTER(C.c_void_p)]
    L.fmd_audio_pcm16_dev.argtypes = [H, C.c_void_p, C.c_void_p]
    L.fmd_rds_dev.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.fmd_get_audio.argtypes = [H, C.c_void_p]
    L.fmd_get_rds_symbols.argtypes = [H, C.c_void_p, C.c_void_p]
    L.fmd_get_rds_bytes.argtypes = [H, C.c_void_p, C.c_int, C.c_void_p]
    L.fmd_rds_bytes_dev.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    L.fmd_get_stream.argtypes = [H, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.fmd_selftest_atan2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.fmd_selftest_atan2_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.fmd_selftest_atan2_table_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.fmd_selftest_atan2_small.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.fmd_selftest_fast_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.fmd_get_spec_stats.argtypes = [H, C.c_void_p, C.c_int]
    L.fmd_design_pll_span.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fmd_profile_enable.argtypes = [H, C.c_int]
    L.fmd_debug_split_front.argtypes = [H, C.c_int]
    L.fmd_debug_set_chain.argtypes = [H, C.c_int]
    L.fmd_debug_pll_adaptive.argtypes = [H, C.c_int, C.c_int]
    L.fmd_debug_extract_pairing.argtypes = [H, C.c_int]
    L.fmd_debug_chain_blocks.argtypes = [H, C.POINTER(C.c_long)]
    L.fmd_profile_read.argtypes = [H, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]
    L.fmd_chan_design.argtypes = [C.c_double, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.fmd_chan_create.argtypes = [C.POINTER(ChanConfig), C.POINTER(C.c_void_p)]
    L.fmd_chan_destroy.argtypes = [C.c_void_p]
    L.fmd_chan_reset.argtypes = [C.c_void_p]
    L.fmd_chan_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    L.fmd_chan_get_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.fmd_chan_process_cf32_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    L.fmd_chan_last_error.restype = C.c_char_p
    L.fmd_chan_last_error.argtypes = [C.c_void_p]
    L.fmd_last_error.restype = C.c_char_p
    L.fmd_last_error.argtypes = [H]
    if L.fmd_output_lifetime_blocks() != FMD_OUTPUT_LIFETIME_BLOCKS:
        raise RuntimeError("libfmdemod.so and capi.py disagree on FMD_OUTPUT_LIFETIME_BLOCKS")
    _lib = L
    return L


def selftest_atan2(y: np.ndarray, x: np.ndarray, table_form: bool = False) -> np.ndarray:
    """The kernels' atan2f evaluated on the device (fmd_selftest_atan2; table_form: the discriminator's variant)."""
    y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(y)
    L = load_library()
    fn = L.fmd_selftest_atan2_table_u8 if table_form == "u8" else (L.fmd_selftest_atan2_table if table_form else L.fmd_selftest_atan2)
    rc = fn(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), y.size)
    if rc != FMD_OK:
        raise FmdError(rc, load_library().fmd_last_error(None).decode())
    return out


def selftest_fast_math(kind: str, a: np.ndarray, b: np.ndarray | None = None) -> np.ndarray:
    """The tolerance mode's primitives on the device: kind in {"atan2", "sin_turns", "cos_turns", "atan2_turns"}."""
    a = np.ascontiguousarray(a, np.float32)
    b = a if b is None else np.ascontiguousarray(b, np.float32)
    out = np.empty_like(a)
    rc = load_library().fmd_selftest_fast_math({"atan2": 0, "sin_turns": 1, "cos_turns": 2, "atan2_turns": 3}[kind], a.ctypes.data_as(C.c_void_p),
                                               b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.size)
    if rc != FMD_OK:
        raise FmdError(rc, load_library().fmd_last_error(None).decode())
    return out


def selftest_atan2_small(y: np.ndarray, x: np.ndarray):
    """The locked-loop short form of atan2f on the device: (values, ok) — values are exact wherever ok is True."""
    y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(y); ok = np.empty(y.size, np.uint8)
    rc = load_library().fmd_selftest_atan2_small(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                 ok.ctypes.data_as(C.c_void_p), y.size)
    if rc != FMD_OK:
        raise FmdError(rc, load_library().fmd_last_error(None).decode())
    return out, ok.astype(bool).reshape(y.shape)


def default_controls() -> Controls:
    c = Controls()
    load_library().fmd_default_controls(C.byref(c))
    return c


def default_config(n_channels: int, fs_baseband: int = 1_024_000) -> "Config":
    """fmd_default_config: 64 ms blocks, the tolerance mode (what a many-station deployment wants; flags = 0 is the bit-exact mode)."""
    cfg = Config()
    rc = load_library().fmd_default_config(C.byref(cfg), n_channels, fs_baseband)
    if rc != FMD_OK:
        raise FmdError(rc, "unsupported configuration")
    return cfg


class BatchDemod:
    """C broadcast-FM demodulators advanced in lock-step on one MI355X."""

    def __init__(self, n_channels: int, block_size: int = 65536, fs_baseband: int = 1_024_000, device: int = -1, keep_taps: bool = False,
                 pipelined: bool = True, pll_kernel: str = "auto", pll_stream_order: bool = False, fast_math: bool = False):
        self.L = load_library()
        self.h = C.c_void_p()
        flags = (FMD_FLAG_KEEP_TAPS if keep_taps else 0) | (0 if pipelined else FMD_FLAG_NO_PIPELINE)
        flags |= {"auto": 0, "time_parallel": FMD_FLAG_PLL_TIME_PARALLEL, "time_parallel8": FMD_FLAG_PLL_TIME_PARALLEL | FMD_FLAG_PLL_K8,
                  "low_work": FMD_FLAG_PLL_LOW_WORK}[pll_kernel]
        flags |= FMD_FLAG_PLL_STREAM_ORDER if pll_stream_order else 0
        flags |= FMD_FLAG_FAST_MATH if fast_math else 0
        cfg = Config(n_channels, block_size, fs_baseband, device, flags)
        rc = self.L.fmd_create(C.byref(cfg), C.byref(self.h))
        if rc != FMD_OK:
            msg = self.L.fmd_last_error(None).decode()
            self.h = None
            raise FmdError(rc, msg or self.L.fmd_status_string(rc).decode())
        self.n_channels, self.block_size, self.fs_baseband = n_channels, block_size, fs_baseband
        r = Rates()
        self._check(self.L.fmd_get_rates(self.h, C.byref(r)))
        self.rates = r
        self.bytes_cap = 16 * (r.n_rds // 256 + 1)

    def close(self):
        if getattr(self, "h", None):
            self.L.fmd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != FMD_OK:
            raise FmdError(rc, self.L.fmd_last_error(self.h).decode() or self.L.fmd_status_string(rc).decode())

    # -- controls (reference GetControls(), broadcast_fm_demod.h:294) --
    def set_controls(self, controls: Controls, channel: int = -1):
        self._check(self.L.fmd_set_controls(self.h, channel, C.byref(controls)))

    def get_coeffs(self, channel: int = 0) -> Coeffs:
        k = Coeffs()
        self._check(self.L.fmd_get_coeffs(self.h, channel, C.byref(k)))
        return k

    def reset(self):
        self._check(self.L.fmd_reset(self.h))

    # -- Process (reference broadcast_fm_demod.cpp:309-328) --
    def process(self, iq, stream=None) -> int:
        """iq: [C, N, 2] float32 or uint8; torch CUDA tensor (device entry point) or numpy array (host entry point).
        Returns the status code (FMD_ERR_SIZE for a dropped block) without raising for size mismatches."""
        is_torch = hasattr(iq, "data_ptr")
        shape = tuple(iq.shape)
        if len(shape) != 3 or shape[2] != 2:
            raise ValueError("iq must be [C, N, 2]")
        if is_torch:
            import torch
            if not iq.is_cuda or not iq.is_contiguous():
                raise ValueError("torch input must be a contiguous CUDA tensor")
            if stream is None:
                stream = torch.cuda.current_stream(iq.device).cuda_stream
            fn = {torch.float32: self.L.fmd_process_cf32_dev, torch.uint8: self.L.fmd_process_u8_dev}[iq.dtype]
            rc = fn(self.h, iq.data_ptr(), shape[0], shape[1], C.c_void_p(stream))
        else:
            a = np.ascontiguousarray(iq)
            fn = {np.dtype(np.float32): self.L.fmd_process_cf32_host, np.dtype(np.uint8): self.L.fmd_process_u8_host}[a.dtype]
            rc = fn(self.h, a.ctypes.data_as(C.c_void_p), shape[0], shape[1])
        if rc not in (FMD_OK, FMD_ERR_SIZE):
            self._check(rc)
        return rc

    def submit(self, iq, ready_stream=None) -> int:
        """fmd_submit_*_dev: like process() for a torch CUDA tensor, but nothing is queued on the caller's streams.  The block is
        read after everything already queued on `ready_stream` (None: the data is in place now); wait_input() tells when the
        buffer may be rewritten.  For hosts that rotate input buffers (bench.py's resident blocks, station_ring.hpp)."""
        import torch
        shape = tuple(iq.shape)
        if len(shape) != 3 or shape[2] != 2 or not iq.is_cuda or not iq.is_contiguous():
            raise ValueError("iq must be a contiguous CUDA tensor [C, N, 2]")
        if hasattr(ready_stream, "cuda_stream"):
            ready_stream = ready_stream.cuda_stream
        fn = {torch.float32: self.L.fmd_submit_cf32_dev, torch.uint8: self.L.fmd_submit_u8_dev}[iq.dtype]
        rc = fn(self.h, iq.data_ptr(), shape[0], shape[1], C.c_void_p(ready_stream) if ready_stream else None)
        if rc not in (FMD_OK, FMD_ERR_SIZE):
            self._check(rc)
        return rc

    def wait_input(self, stream=None):
        """Make `stream` wait (on the device) until the newest submitted block's input buffer has been read."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self._check(self.L.fmd_wait_input(self.h, C.c_void_p(stream)))

    def synchronize(self):
        self._check(self.L.fmd_synchronize(self.h))

    def set_chain(self, on: bool) -> None:
        """fmd_debug_set_chain: False keeps the three-launch form of the tolerance mode's steady blocks (the parity A/B of k_chain)."""
        self._check(self.L.fmd_debug_set_chain(self.h, 1 if on else 0))

    def pll_adaptive(self, k16_max_channels: int, time_parallel_max_channels: int = 7168) -> None:
        """fmd_debug_pll_adaptive: exact mode — the batch sizes from which the pilot-PLL kernel (its lane count; low-work or time-parallel) is picked
        by what is out of lock (small values let a small batch exercise the switches)."""
        self._check(self.L.fmd_debug_pll_adaptive(self.h, int(k16_max_channels), int(time_parallel_max_channels)))

    def set_extract_pairing(self, mode: int) -> None:
        """fmd_debug_extract_pairing: 0 auto, 1 wherever possible, 2 never (k_extract_bp with two stations per workgroup)."""
        self._check(self.L.fmd_debug_extract_pairing(self.h, int(mode)))

    def chain_blocks(self) -> int:
        n = C.c_long(0)
        self._check(self.L.fmd_debug_chain_blocks(self.h, C.byref(n)))
        return int(n.value)

    def set_output_lag(self, on: bool) -> None:
        """fmd_set_output_lag: with on=True the device-side output calls (wait_outputs, release_outputs, audio_tensor, audio_pcm16_into,
        ...) refer to the newest block whose output stages are QUEUED — behind submit() of block k that is block k - 1 in the
        tolerance mode — and never force a put-off stage."""
        self._check(self.L.fmd_set_output_lag(self.h, 1 if on else 0))

    def outputs_block(self) -> int:
        """fmd_outputs_block: index (0 = first since create / reset) of the block the device-side output calls refer to, -1 = none yet."""
        b = C.c_long(-1)
        self._check(self.L.fmd_outputs_block(self.h, C.byref(b)))
        return int(b.value)

    def wait_outputs(self, stream=None):
        """Make `stream` (torch stream or raw handle; default: torch current stream) wait for the newest block's outputs."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self._check(self.L.fmd_wait_outputs(self.h, C.c_void_p(stream)))

    def release_outputs(self, stream=None):
        """Tell the library that everything queued on `stream` so far reads the newest block's output views: their buffers are
        not reused before that work has finished (fmd_release_outputs; device-side ordering, the host never blocks)."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self._check(self.L.fmd_release_outputs(self.h, C.c_void_p(stream)))

    # -- per-channel state snapshot / restore --
    def get_state(self, channel: int) -> bytes:
        n = self.L.fmd_state_size(self.h)
        buf = C.create_string_buffer(n)
        self._check(self.L.fmd_get_state(self.h, channel, buf, n))
        return buf.raw

    def set_state(self, channel: int, blob: bytes):
        self._check(self.L.fmd_set_state(self.h, channel, blob, len(blob)))

    # -- outputs --
    def audio(self) -> np.ndarray:
        out = np.empty((self.n_channels, self.rates.n_audio, 2), np.float32)
        self._check(self.L.fmd_get_audio(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def rds_symbols(self) -> tuple[np.ndarray, np.ndarray]:
        syms = np.empty((self.n_channels, self.rates.n_rds), np.float32)
        counts = np.empty(self.n_channels, np.int32)
        self._check(self.L.fmd_get_rds_symbols(self.h, syms.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p)))
        return syms, counts

    def rds_bytes(self) -> tuple[np.ndarray, np.ndarray]:
        b = np.zeros((self.n_channels, self.bytes_cap), np.uint8)
        counts = np.empty(self.n_channels, np.int32)
        self._check(self.L.fmd_get_rds_bytes(self.h, b.ctypes.data_as(C.c_void_p), self.bytes_cap, counts.ctypes.data_as(C.c_void_p)))
        return b, counts

    def stream(self, name: str) -> np.ndarray:
        n = C.c_size_t(0)
        probe = np.empty(1, np.float32)
        self.L.fmd_get_stream(self.h, name.encode(), probe.ctypes.data_as(C.c_void_p), 0, C.byref(n))
        if n.value == 0:
            self._check(self.L.fmd_get_stream(self.h, name.encode(), probe.ctypes.data_as(C.c_void_p), 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        self._check(self.L.fmd_get_stream(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
        return out.reshape(self.n_channels, -1)

    def profile(self, on):
        """False/0: off; True/1: bracket every kernel of every block; 2: the dominant kernel every block, the rest every 4th;
        3: every kernel of every 4th block, plus the dominant kernel of the block behind it."""
        self._check(self.L.fmd_profile_enable(self.h, int(on)))

    def spec_stats(self, reset: bool = False) -> dict:
        """Speculation counters of the pilot PLL kernel (fmd_get_spec_stats) since creation / the last reset."""
        a = np.zeros(8, np.uint64)
        self._check(self.L.fmd_get_spec_stats(self.h, a.ctypes.data_as(C.c_void_p), 1 if reset else 0))
        out = {"pll": {"chunks": int(a[0]), "serial_chunks": int(a[1]), "exact_spans": int(a[2]), "spans": int(a[3]), "samples": int(a[4]),
                       "samples_per_span": float(a[4]) / float(a[3]) if a[3] else 0.0, "sequence_spans": int(a[5])}}
        if a[7]:
            out["pll_clock_mhz"] = float(a[6]) / float(a[7]) * 100.0
        return out

    def profile_read(self) -> dict:
        """{kernel name: (total ms, launches)} since the last read (HIP events on the processing stream)."""
        arr = (KernelTime * 16)()
        n = C.c_int(0)
        self._check(self.L.fmd_profile_read(self.h, arr, 16, C.byref(n)))
        return {arr[i].name.decode(): (arr[i].total_ms, arr[i].launches) for i in range(n.value)}

    def audio_pcm16_into(self, out, stream=None):
        """Convert the newest block's audio to the reference scraper's 16-bit PCM frames into `out` (torch int16 CUDA tensor
        [C, n_audio, 2]) on `stream` (default: torch current stream), ordered behind the block's outputs."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        assert out.dtype == torch.int16 and out.is_contiguous() and out.numel() == self.n_channels * self.rates.n_audio * 2
        self._check(self.L.fmd_audio_pcm16_dev(self.h, C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def audio_tensor(self):
        """Zero-copy torch view of the newest block's device audio buffer [C, n_audio, 2] (the library alternates
        between its pipeline slots: call again after each process(); contents are complete after wait_outputs/synchronize and stay
        valid while at most FMD_OUTPUT_LIFETIME_BLOCKS = 5 further blocks have been submitted — or longer for a consumer that
        calls release_outputs())."""
        import torch
        p = C.c_void_p()
        self._check(self.L.fmd_audio_dev(self.h, C.byref(p)))
        n = self.n_channels * self.rates.n_audio * 2

        class _Arr:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (p.value, False), "version": 2}
        return torch.as_tensor(_Arr(), device="cuda").view(self.n_channels, self.rates.n_audio, 2)

    def rds_bytes_tensors(self):
        """Zero-copy torch views of the newest block's RDS byte buffers: (bytes [C, cap] uint8, counts [C] int32) — fmd_rds_bytes_dev; same
        lifetime rule as audio_tensor()."""
        import torch
        pb, pc, cap = C.c_void_p(), C.c_void_p(), C.c_int(0)
        self._check(self.L.fmd_rds_bytes_dev(self.h, C.byref(pb), C.byref(pc), C.byref(cap)))
        nC = self.n_channels

        class _B:
            __cuda_array_interface__ = {"shape": (nC * cap.value,), "typestr": "|u1", "data": (pb.value, False), "version": 2}

        class _C:
            __cuda_array_interface__ = {"shape": (nC,), "typestr": "<i4", "data": (pc.value, False), "version": 2}
        return torch.as_tensor(_B(), device="cuda").view(nC, cap.value), torch.as_tensor(_C(), device="cuda")


def chan_design(fs_in: float, fs_out: float, taps_per_phase: int = 640):
    """Host-only prototype design of the wideband channeliser: (taps [T, L] float32, L, M).  Needs no GPU."""
    lib = load_library()
    L, M = C.c_int(0), C.c_int(0)
    rc = lib.fmd_chan_design(fs_in, fs_out, taps_per_phase, None, C.byref(L), C.byref(M))
    if rc != FMD_OK:
        raise FmdError(rc, "unsupported channeliser rates")
    taps = np.empty((taps_per_phase, L.value), np.float32)
    lib.fmd_chan_design(fs_in, fs_out, taps_per_phase, taps.ctypes.data_as(C.c_void_p), None, None)
    return taps, L.value, M.value


class Channelizer:
    """Wideband capture -> [C][n_out] cf32 stations at fs_out on the GPU (fmd_chan_*); feeds BatchDemod.process directly."""

    def __init__(self, fs_in: float, center_hz, fs_out: float = 256_000.0, max_input_samples: int = 640_000, taps_per_phase: int = 0, device: int = -1):
        self.L = load_library()
        self.centers = np.ascontiguousarray(center_hz, np.float64)
        cfg = ChanConfig(fs_in, fs_out, int(self.centers.size), self.centers.ctypes.data_as(C.POINTER(C.c_double)), taps_per_phase, max_input_samples, device)
        self.h = C.c_void_p()
        rc = self.L.fmd_chan_create(C.byref(cfg), C.byref(self.h))
        if rc != FMD_OK:
            raise FmdError(rc, self.L.fmd_chan_last_error(None).decode())
        l, m, t, c = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.L.fmd_chan_info(self.h, C.byref(l), C.byref(m), C.byref(t), C.byref(c))
        self.interp, self.decim, self.taps_per_phase, self.n_stations = l.value, m.value, t.value, c.value

    def taps(self) -> np.ndarray:
        a = np.empty((self.taps_per_phase, self.interp), np.float32)
        rc = self.L.fmd_chan_get_taps(self.h, a.ctypes.data_as(C.c_void_p), a.size)
        if rc != FMD_OK:
            raise FmdError(rc, "fmd_chan_get_taps")
        return a

    def process(self, wide, out=None, stream=None):
        """wide: contiguous CUDA float32 tensor [n_in, 2]; returns a CUDA tensor [C, n_out, 2] (asynchronous on the stream)."""
        import torch
        if not (wide.is_cuda and wide.is_contiguous() and wide.dtype == torch.float32 and wide.dim() == 2 and wide.shape[1] == 2):
            raise ValueError("wide must be a contiguous CUDA float32 tensor [n_in, 2]")
        n_in = int(wide.shape[0])
        n_out = n_in * self.interp // self.decim
        if out is None:
            out = torch.empty((self.n_stations, n_out, 2), dtype=torch.float32, device=wide.device)
        if stream is None:
            stream = torch.cuda.current_stream(wide.device).cuda_stream
        got = C.c_size_t(0)
        rc = self.L.fmd_chan_process_cf32_dev(self.h, wide.data_ptr(), n_in, out.data_ptr(), int(out.shape[1]), C.byref(got), C.c_void_p(stream))
        if rc != FMD_OK:
            raise FmdError(rc, self.L.fmd_chan_last_error(self.h).decode())
        return out[:, :got.value]

    def reset(self):
        rc = self.L.fmd_chan_reset(self.h)
        if rc != FMD_OK:
            raise FmdError(rc, self.L.fmd_chan_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.fmd_chan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
