"""fm-radio_amd — MI355X-native batched broadcast-FM demodulator.

The product is `csrc/libfmdemod.so` (hand-written HIP kernels behind the C ABI in include/fmdemod.h).
This package is the thin Python plumbing around it: building the library, a ctypes binding, and
helpers that hand torch device tensors (device memory, streams, torch.distributed) to the C ABI.
Nothing here computes anything on the CPU: if the library or a gfx950 device is missing, calls raise.

The directory name contains a hyphen, so import it through `fmradio_loader.load()` (repo root) or
`importlib` with the module name `fm_radio_amd`.
"""
from .capi import (  # noqa: F401
    FMD_FLAG_FAST_MATH,
    FMD_AUDIO_LMR, FMD_AUDIO_LPR, FMD_AUDIO_STEREO, FMD_FLAG_KEEP_TAPS, FMD_FLAG_NO_PIPELINE, BatchDemod, Coeffs, Config, Controls, FmdError,
    Channelizer, Rates, build_library, chan_design, declared_symbols, default_config, default_controls, lib_path, load_library, selftest_atan2, selftest_atan2_small, selftest_fast_math,
)
from .sharding import AudioGather, channel_range, padded_shard  # noqa: F401,E402
