#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference (oracle/_ref/fm_ref_dump).

Run in the build container only (needs /root/reference mounted so `make -C oracle ref` works):

    python tests/golden/make_golden.py

Outputs (data only — inputs and the reference's outputs):
  chain_b16384.npz   6 blocks of 16384 u8 IQ + every stream the reference exposes
  chain_cf32_b8192.npz  4 blocks of 8192 cf32 IQ through Broadcast_FM_Demod::Process directly
  long_b65536.npz    2.6 s run (40 blocks of 65536): capture regenerated from its seed (sha256 pinned),
                     RDS symbols/bytes in full, audio of 3 blocks + sha256 of all audio
  prims.npz          per-primitive input/output vectors (filters with history carry, AGC, discriminator, ...)
  taps.npz           every designed coefficient set used by the chain
"""
from __future__ import annotations

import hashlib
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "oracle"))

import oraclelib as O  # noqa: E402
import synth  # noqa: E402

LONG_SEED, LONG_BLOCKS, LONG_BS = 4321, 40, 65536


def long_capture() -> np.ndarray:
    c = synth.fm_capture(LONG_BLOCKS * LONG_BS, seed=LONG_SEED)
    return synth.to_u8(c["iq"]), c["groups"]


def prim_inputs(rng: np.random.Generator) -> dict:
    n = 4096
    t = np.arange(n)
    x_c = (np.exp(2j * np.pi * 0.0371 * t) * (1.0 + 0.3 * np.sin(2 * np.pi * 0.003 * t))
           + 0.2 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
    x_c = np.stack([x_c.real, x_c.imag], axis=1).astype(np.float32)
    x_r = (np.sin(2 * np.pi * 0.0113 * t) + 0.1 * rng.standard_normal(n)).astype(np.float32)
    grid = np.linspace(-0.5, 0.5, 1001).astype(np.float32)
    dt = (((np.arange(1025) * 0.148437) + 0.5) % 1.0 - 0.5).astype(np.float32)
    return {"x_c": x_c, "x_r": x_r, "grid": grid, "dt": dt}


def main() -> None:
    assert O.have_ref(), "build oracle/_ref first: make -C oracle ref"
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        # 1. short chain, u8 ingest, block 16384
        cap = synth.to_u8(synth.fm_capture(6 * 16384, seed=1234)["iq"])
        ref = O.run_ref_chain(cap, td / "a", 16384)
        np.savez_compressed(HERE / "chain_b16384.npz", capture=cap, **ref)
        # 2. cf32 boundary, block 8192
        capf = synth.to_cf32(synth.fm_capture(4 * 8192, seed=99)["iq"])
        ref = O.run_ref_chain(capf, td / "b", 8192, u8=False)
        np.savez_compressed(HERE / "chain_cf32_b8192.npz", capture=capf, **ref)
        # 3. long run
        capl, groups = long_capture()
        ref = O.run_ref_chain(capl, td / "c", LONG_BS)
        audio = ref["audio"].reshape(LONG_BLOCKS, -1)
        np.savez_compressed(
            HERE / "long_b65536.npz",
            seed=LONG_SEED, n_blocks=LONG_BLOCKS, block_size=LONG_BS,
            capture_sha256=hashlib.sha256(capl.tobytes()).hexdigest(),
            audio_sha256=hashlib.sha256(ref["audio"].tobytes()).hexdigest(),
            audio_blocks=np.array([0, 20, 39]), audio=audio[[0, 20, 39]],
            rds_sym=ref["rds_sym"], rds_count=ref["rds_count"], rds_bytes=ref["rds_bytes"],
            lmr_phase=ref["lmr_phase"], groups=np.array(groups[:64], dtype=np.uint16),
        )
        # 4. primitives + taps
        pin = prim_inputs(np.random.default_rng(7))
        ind, outd = td / "pin", td / "pout"
        ind.mkdir(); outd.mkdir()
        pin["x_c"].tofile(ind / "x_c.cf32"); pin["x_r"].tofile(ind / "x_r.f32")
        pin["grid"].tofile(ind / "grid.f32"); pin["dt"].tofile(ind / "dt.f32")
        subprocess.run([str(O.REF_DUMP), "prims", str(ind), str(outd)], check=True)
        outs = {p.name.replace(".", "_"): np.fromfile(p, dtype=np.float32) for p in sorted(outd.iterdir()) if p.name != "taps.f32"}
        np.savez_compressed(HERE / "prims.npz", **pin, **outs)
        np.savez_compressed(HERE / "taps.npz", taps=np.fromfile(outd / "taps.f32", dtype=np.float32))
    for p in sorted(HERE.glob("*.npz")):
        print(p.name, p.stat().st_size)


if __name__ == "__main__":
    main()
