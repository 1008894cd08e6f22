"""CPU-only checks of the product's boundary: the C-ABI library builds, loads and exports every symbol
include/fmdemod.h declares; without a GPU it refuses to work (no CPU fallback); host-side filter design and
the device math header (compiled for the host) agree with the oracle / libm."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import fmradio_loader
import oraclelib as O

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def pkg():
    p = fmradio_loader.load()
    p.build_library()
    return p


def test_library_exports_declared_abi(pkg):
    lib = pkg.load_library()
    names = pkg.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fmdemod.h but not exported"
    assert lib.fmd_api_version() == 3
    assert lib.fmd_status_string(-4).decode().startswith("no gfx950")
    assert "fmd_reset" in lib.fmd_status_string(-6).decode()


def test_debug_hooks_live_in_their_own_header(pkg):
    """The drop-in boundary (include/fmdemod.h) holds only entry points with a counterpart in the reference's demodulator
    API; self-tests and profiling hooks are declared in include/fmdemod_debug.h."""
    product = set(pkg.declared_symbols(debug=False))
    everything = set(pkg.declared_symbols())
    hooks = everything - product
    assert hooks == {"fmd_selftest_atan2", "fmd_selftest_atan2_table", "fmd_selftest_atan2_table_u8", "fmd_selftest_atan2_small", "fmd_selftest_fast_math",
                     "fmd_get_spec_stats", "fmd_profile_enable", "fmd_profile_read", "fmd_design_pll_span", "fmd_design_pll_sparse", "fmd_debug_split_front",
                     "fmd_design_extract_bp", "fmd_design_wrap_tie", "fmd_debug_set_chain", "fmd_debug_chain_blocks", "fmd_debug_extract_pairing", "fmd_debug_pll_adaptive"}
    assert {"fmd_submit_cf32_dev", "fmd_submit_u8_dev", "fmd_wait_input", "fmd_set_output_lag", "fmd_outputs_block", "fmd_outputs_epoch"} <= product
    assert {"fmd_release_outputs", "fmd_get_state", "fmd_set_state", "fmd_state_size", "fmd_output_lifetime_blocks"} <= product
    # both headers compile as plain C
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = Path(td) / "c.c"
        src.write_text('#include "fmdemod.h"\n#include "fmdemod_debug.h"\nint main(void) { return FMD_OUTPUT_LIFETIME_BLOCKS == 5 ? 0 : 1; }\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(Path(td) / "c")], check=True)
        subprocess.run([str(Path(td) / "c")], check=True)


def test_one_output_lifetime_contract(pkg):
    """The number of further blocks an output view survives is stated once (FMD_OUTPUT_LIFETIME_BLOCKS) and the header, the
    library, the Python binding and INTEGRATION.md agree on it."""
    lib = pkg.load_library()
    n = lib.fmd_output_lifetime_blocks()
    from fm_radio_amd import capi
    assert n == capi.FMD_OUTPUT_LIFETIME_BLOCKS == 5
    hdr = (ROOT / "include" / "fmdemod.h").read_text()
    assert f"#define FMD_OUTPUT_LIFETIME_BLOCKS {n}" in hdr
    assert "second\n * next" not in hdr and "three blocks in" not in hdr      # the stale round-1 wordings
    integ = (ROOT / "INTEGRATION.md").read_text()
    assert "FMD_OUTPUT_LIFETIME_BLOCKS" in integ and "five more blocks" not in integ


def test_library_does_not_touch_the_environment(pkg):
    """Loading libfmdemod.so must not mutate the host process's environment (round 1 had a constructor calling setenv)."""
    code = ("import ctypes, os; os.environ.pop('GPU_MAX_HW_QUEUES', None); "
            f"ctypes.CDLL(r'{pkg.lib_path()}'); print(os.environ.get('GPU_MAX_HW_QUEUES'), ctypes.CDLL(None).getenv(b'GPU_MAX_HW_QUEUES'))")
    import sys
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["None", "0"], out


def test_no_gpu_means_loud_failure(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert pkg.load_library().fmd_device_count() == 0
    with pytest.raises(pkg.FmdError) as e:
        pkg.BatchDemod(1, 8192, 1_024_000)
    assert e.value.status == -4  # FMD_ERR_NO_DEVICE


def test_bad_configs_rejected(pkg):
    lib = pkg.load_library()
    h = C.c_void_p()
    for cfg in (pkg.Config(0, 8192, 1_024_000, -1, 0), pkg.Config(1, 1000, 1_024_000, -1, 0), pkg.Config(1, 8192, 48000, -1, 0),
                pkg.Config(1, 4096, 2_048_000, -1, 0), pkg.Config(1, 8192, 1_024_000, -1, 1 << 20)):   # last: an unknown flag bit
        assert lib.fmd_create(C.byref(cfg), C.byref(h)) == -1
        assert not h.value


def test_host_designer_matches_oracle(pkg, tmp_path):
    """fmd_design.cpp (product, host side) == oracle designs bit for bit (IEEE variant of the pilot gain)."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "fmd_design.h"
#include <stdio.h>
#include <initializer_list>
int main() {
    for (int fs : {256000, 1024000, 2048000}) {
        fmd_controls c; c.audio_out = 2; c.audio_stereo_mix_factor = 1; c.use_deemphasis = 0; c.deemphasis_tus = 75; c.lpr_cutoff_hz = 12000; c.lmr_cutoff_hz = 9000;
        fmd_coeffs k; fmd::design_all(&k, fs, &c);
        fwrite(&k, sizeof(k), 1, stdout);
    }
    return 0;
}''')
    exe = tmp_path / "t"
    csrc = ROOT / "fm-radio_amd" / "csrc"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{csrc}", f"-I{ROOT / 'include'}", str(src), str(csrc / "fmd_design.cpp"), "-o", str(exe)], check=True)
    raw = subprocess.run([str(exe)], check=True, capture_output=True).stdout
    sz = C.sizeof(O.Coeffs)
    assert len(raw) == 3 * sz
    ctl = O.default_controls()
    ctl.deemphasis_tus, ctl.lpr_cutoff_hz, ctl.lmr_cutoff_hz = 75, 12000, 9000
    for i, fs in enumerate((256000, 1024000, 2048000)):
        got = O.Coeffs.from_buffer_copy(raw[i * sz:(i + 1) * sz])
        want = O.design(fs, ctl, rsqrt_mode=0)
        assert bytes(got) == bytes(want), f"fs={fs}"


def test_device_math_atan2_matches_libm(pkg, tmp_path):
    """fmd_math.h's atan2f (the function the kernels use) against the host libm the reference links: bit-identical."""
    src = tmp_path / "a.cpp"
    src.write_text(r'''
#include "fmd_math.h"
#include <stdio.h>
#include <random>
int main() {
    std::mt19937_64 rng(99); long bad = 0, n = 0;
    auto chk = [&](float y, float x) { float a = atan2f(y, x), b = fmd::fmd_atan2f(y, x); uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
        n++; if (ua != ub && !(a != a && b != b)) bad++; };
    for (long i = 0; i < 4000000; i++) { uint64_t r = rng(); uint32_t a = (uint32_t)r, b = (uint32_t)(r >> 32); float y, x; memcpy(&y, &a, 4); memcpy(&x, &b, 4); chk(y, x); }
    std::uniform_real_distribution<float> U(-200.f, 200.f), V(-1.5f, 1.5f);
    for (long i = 0; i < 4000000; i++) chk(U(rng), U(rng));
    for (long i = 0; i < 4000000; i++) chk(V(rng), V(rng));
    float sp[] = {0.f, -0.f, 1.f, -1.f, INFINITY, -INFINITY, NAN, 1e-45f, -1e-45f, 1e38f, -1e38f, 1e-38f, 0.4375f, 0.6875f, 1.1875f, 2.4375f};
    for (float y : sp) for (float x : sp) chk(y, x);
    printf("%ld %ld\n", n, bad); return 0;
}''')
    exe = tmp_path / "a"
    csrc = ROOT / "fm-radio_amd" / "csrc"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{csrc}", str(src), "-o", str(exe)], check=True)
    n, bad = map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    assert n > 12_000_000 and bad == 0


def test_default_config_is_the_tolerance_mode_with_64_ms_blocks(pkg):
    """fmd_default_config: the mode a band scan wants is what a caller gets without knowing the flags; flags = 0 stays the
    bit-exact mode for parity harnesses (reference block: 65536 samples at 1.024 MSa/s, broadcast_fm_demod.cpp:62-77)."""
    for fs, bs in ((256_000, 16384), (1_024_000, 65536), (2_048_000, 131072)):
        cfg = pkg.default_config(40, fs)
        assert (cfg.n_channels, cfg.block_size, cfg.fs_baseband, cfg.device) == (40, bs, fs, -1)
        assert cfg.flags == pkg.FMD_FLAG_FAST_MATH
    with pytest.raises(pkg.FmdError):
        pkg.default_config(4, 48_000)
