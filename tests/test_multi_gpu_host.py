"""The multi-GPU C++ host (fm-radio_amd/host/multi_gpu_host.hpp) and the RCCL output gather behind include/fmdemod_gather.h
(reference anchor: one demodulator per station wired to audio AND RDS observers, src/app.cpp:19-34).

CPU: libfmdgather.so builds, exports what its header declares, and the host header + driver compile against the C ABIs alone.
GPU (one MI355X here): two ranks sharing the device (threads, shard bookkeeping, copy hand-over, back-pressure, f32 and PCM16, RDS
bytes) and one rank whose shard travels through ncclSend / ncclRecv to the self peer (RCCL refuses two ranks on one device:
tools/rccl_probe.cpp); with >= 2 devices the same driver puts every rank on its own GPU and everything crosses RCCL."""
import ctypes as C
import json
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

import synth

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "fm-radio_amd" / "csrc"


def build_driver(tmp_path) -> Path:
    exe = tmp_path / "multi_gpu_main"
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}", f"-I{ROOT / 'fm-radio_amd' / 'host'}",
                    str(ROOT / "tests" / "cpp" / "multi_gpu_main.cpp"), f"-L{CSRC}", "-lfmdgather", "-lfmdemod", "-L/opt/rocm/lib", "-lamdhip64",
                    "-lpthread", f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    return exe


def test_gather_library_builds_and_exports_its_header(tmp_path):
    import fmradio_loader
    fmradio_loader.load().build_library()
    lib = C.CDLL(str(CSRC / "libfmdgather.so"))
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "fmdemod_gather.h").read_text(), flags=re.S)
    names = sorted(set(re.findall(r"\b(fmd_gather_[a-z0-9_]+)\s*\(", text)))
    assert names == ["fmd_gather_abort", "fmd_gather_collector", "fmd_gather_create", "fmd_gather_destroy", "fmd_gather_last_error", "fmd_gather_remote_bytes_per_block",
                     "fmd_gather_submit", "fmd_gather_wait"]
    for n in names:
        assert hasattr(lib, n), n
    assert build_driver(tmp_path).exists()          # the host header and its driver: plain C++ against the two C ABIs
    # the header is plain C
    src = tmp_path / "c.c"
    src.write_text('#include "fmdemod_gather.h"\nint main(void) { return FMD_GATHER_PCM16 == 1 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(tmp_path / "c")], check=True)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,fmt,loopback,fast,rotate", [(2, "pcm16", 0, True, False), (2, "f32", 0, False, False), (1, "pcm16", 1, True, False), (1, "f32", 1, False, False),
                                                             (2, "pcm16", 0, True, True), (3, "f32", 0, True, True)])
def test_multi_gpu_host_gathers_audio_and_rds_bytes(tmp_path, ranks, fmt, loopback, fast, rotate):
    """With as many GPUs as ranks the driver puts every rank on its own device (everything crosses RCCL); `rotate`: the collector moves
    from rank to rank block by block (FMD_GATHER_ROTATE) — from GPU to GPU with one rank per device, between the ranks' buffer sets on the
    one device of a one-GPU box (every hand-over a copy: the rotation's bookkeeping runs either way)."""
    import fmradio_loader
    import torch
    fmradio_loader.load().load_library()
    exe = build_driver(tmp_path)
    c_local, bs, fs, nb = 6, 16384, 256_000, 10                   # 0.64 s: the Manchester decoder has handed on bytes by then
    base = np.stack([synth.to_u8(synth.fm_capture(nb * bs, fs=float(fs), seed=5600, channel=c)["iq"]) for c in range(4)])
    caps = base[np.arange(ranks * c_local) % 4]
    f = tmp_path / "caps.u8"
    np.ascontiguousarray(caps).tofile(f)
    r = subprocess.run([str(exe), str(f), str(ranks), str(c_local), str(bs), str(fs), str(nb), fmt, str(loopback)] + (["fast"] if fast else []) + (["rotate"] if rotate else []),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["lockstep_mismatches"] == 0 and info["pipelined_mismatches"] == 0
    assert info["rds_bytes_gathered"] >= 16 * ranks * c_local     # the RDS part of the comparison is not vacuous


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,loopback,rotate", [(2, 0, True), (1, 1, False)])
def test_a_failed_rank_turns_into_an_error_not_a_hang(tmp_path, ranks, loopback, rotate):
    """a rank whose fmd_submit fails never posts its part of the gather.  Collect() must throw that rank's error, and the
    host's destructor — fmd_gather_abort, join, fmd_gather_destroy, which aborts the communicators BEFORE it drains the rank streams —
    must return (the driver runs under a timeout)."""
    import fmradio_loader
    fmradio_loader.load().load_library()
    exe = build_driver(tmp_path)
    c_local, bs, fs, nb = 4, 16384, 256_000, 6
    base = np.stack([synth.to_u8(synth.fm_capture(nb * bs, fs=float(fs), seed=5700, channel=c)["iq"]) for c in range(2)])
    caps = base[np.arange(ranks * c_local) % 2]
    f = tmp_path / "caps.u8"
    np.ascontiguousarray(caps).tofile(f)
    r = subprocess.run([str(exe), str(f), str(ranks), str(c_local), str(bs), str(fs), str(nb), "pcm16", str(loopback), "fast", "failrank"] + (["rotate"] if rotate else []),
                       capture_output=True, text=True, timeout=120)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["failed_rank_threw"] == 1 and info["failed_rank_run_ms"] < 60_000, (info, r.stderr[-1000:])


@pytest.mark.gpu
@pytest.mark.parametrize("loopback", [0, 1])
def test_gather_refuses_a_repeated_and_a_skipped_block_and_follows_a_reset(tmp_path, loopback):
    """fmd_gather_submit tells a handle whose numbering RESTARTED (fmd_reset: fmd_outputs_epoch changes — re-based, the
    gather goes on) from a second submit without a new block and from a skipped block (both FMD_ERR_ARG; round 5 inferred the restart from
    "block <= last" and waved the first through; a lagged handle's "block before" is the same refusal: tests/test_gather_plan_cpu.py).  Through the C ABI, one rank, copy hand-over and
    RCCL loop-back; the bookkeeping alone runs without a GPU in tests/test_gather_plan_cpu.py."""
    import fmradio_loader
    fmradio_loader.load().load_library()
    exe = build_driver(tmp_path)
    c_local, bs, fs, nb = 4, 16384, 256_000, 4
    base = np.stack([synth.to_u8(synth.fm_capture(nb * bs, fs=float(fs), seed=5800, channel=c)["iq"]) for c in range(2)])
    caps = base[np.arange(c_local) % 2]
    f = tmp_path / "caps.u8"
    np.ascontiguousarray(caps).tofile(f)
    r = subprocess.run([str(exe), str(f), "1", str(c_local), str(bs), str(fs), str(nb), "pcm16", str(loopback), "fast", "misuse"], capture_output=True, text=True, timeout=300)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["misuse_handled"] == 1, (info, r.stderr[-1500:])
    assert r.returncode == 0, r.stderr[-1500:]
