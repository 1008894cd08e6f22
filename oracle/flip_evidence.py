#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only; needs /root/reference): how far two legitimate builds of the REFERENCE itself are
apart on the L-R path, and how often its L-R phase tracker's sign decision (reference broadcast_fm_demod.cpp:500-510) falls
differently between them.

    make -C oracle ref ref-scalar          # _ref/fm_ref_dump (its gcc preset: -O2 -ffast-math, AVX2+FMA), _ref/fm_ref_dump_scalar (-fno-fast-math, SSE2, no FMA)
    python3 oracle/flip_evidence.py [stations] [seconds] > profiles/round4/reference_flip_evidence.json

Each estimate of a block is +-pi/2 - atan2(im, re) by the SIGN of an L-R sample; where that sample is within the arithmetic difference
of the two builds of zero, the estimates are pi apart and the blocks' offsets 0.1 pi / n_est.  The tolerance mode of the GPU library
meets the same discontinuity against the oracle (tests/test_gpu_fast.py lmr_audio_excess, tests/test_gpu_long.py).
"""
import json
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import synth  # noqa: E402

BS = 65536


def run(exe: Path, cap: Path, td: Path) -> dict:
    out = td / ("out_" + exe.name)
    out.mkdir()
    subprocess.run([str(exe), "chain", str(cap), str(out), str(BS)], check=True, stderr=subprocess.DEVNULL)
    res = {k: np.fromfile(out / f, np.float32) for k, f in (("lmr", "lmr.f32"), ("audio", "audio.f32"), ("lmr_phase", "lmr_phase.f32"), ("lpr", "lpr.f32"))}
    res["rds_bytes"] = np.fromfile(out / "rds_bytes.u8", np.uint8)
    res["rds_sym"] = np.fromfile(out / "rds_sym.f32", np.float32)
    res["rds_count"] = np.fromfile(out / "rds_count.i32", np.int32)
    shutil.rmtree(out)
    return res


def main() -> None:
    n_st = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    nb = int(secs * 1.024e6) // BS
    a_exe, b_exe = ROOT / "_ref" / "fm_ref_dump", ROOT / "_ref" / "fm_ref_dump_scalar"
    rows = []
    for c in range(n_st):
        with tempfile.TemporaryDirectory() as td_:
            td = Path(td_)
            cap = td / "cap.u8"
            synth.to_u8(synth.fm_capture(nb * BS, seed=2300, channel=c)["iq"]).tofile(cap)
            a, b = run(a_exe, cap, td), run(b_exe, cap, td)
        n_a = BS // 32
        dph = (a["lmr_phase"][:nb].astype(np.float64) - b["lmr_phase"][:nb])
        jumps = np.abs(np.diff(np.concatenate([[0.0], dph])))
        n_est = (n_a + 9) // 10
        flips = int(np.sum(jumps > 0.5 * 0.1 * np.pi / n_est))
        dl = (a["lmr"].astype(np.float64) - b["lmr"]).reshape(nb, -1)
        da = (a["audio"].astype(np.float64) - b["audio"]).reshape(nb, -1)
        per_l = np.sqrt((dl ** 2).mean(axis=1))
        # soft RDS symbols (reference OnRDSOut, broadcast_fm_demod.cpp:327): where the two builds' symbol clocks agree (same count in every
        # block from some block on, at least 0.5 s in), the differences of the symbol VALUES — a zero-crossing or clock-wrap decision of the
        # synchroniser (bpsk_synchroniser.cpp:159-183) that falls on the other side moves a few symbols by ~0.1-0.3, their signs (the bits) stay
        ca, cb = a["rds_count"][:nb], b["rds_count"][:nb]
        agree_from = 0
        for blk in range(nb - 1, -1, -1):
            if ca[blk] != cb[blk]:
                agree_from = blk + 1
                break
        first = max(agree_from, 8)
        sym = None
        if first < nb:
            la, lb = int(ca[:first].sum()), int(cb[:first].sum())
            m = int(ca[first:].sum())
            ds = np.abs(a["rds_sym"][la:la + m].astype(np.float64) - b["rds_sym"][lb:lb + m])
            sym = {"compared_from_block": first, "symbols": m, "median": float(np.median(ds)), "p99": float(np.percentile(ds, 99)), "rms": float(np.sqrt(np.mean(ds ** 2))),
                   "max": float(ds.max()), "symbols_over_1e-2": int(np.sum(ds > 1e-2)), "rms_without_those": float(np.sqrt(np.mean(ds[ds <= 1e-2] ** 2)))}
        rows.append({"station": c, "blocks": nb, "rds_soft_symbols": sym, "flipped_estimates": flips, "lmr_rms_whole_run": float(np.sqrt((dl ** 2).mean())),
                     "audio_rms_whole_run": float(np.sqrt((da ** 2).mean())), "lpr_rms_whole_run": float(np.sqrt(np.mean((a["lpr"].astype(np.float64) - b["lpr"]) ** 2))),
                     "lmr_rms_median_block": float(np.median(per_l)), "lmr_rms_worst_block": float(per_l.max()),
                     "rds_bytes_identical": bool(np.array_equal(a["rds_bytes"], b["rds_bytes"]))})
        print(rows[-1], file=sys.stderr)
    tot_blocks = sum(r["blocks"] for r in rows)
    tot_flips = sum(r["flipped_estimates"] for r in rows)
    print(json.dumps({
        "what": "two builds of the reference itself on the same synthetic 1.024 MSa/s u8 captures (block 65536): its gcc preset build "
                "(-O2 -ffast-math -march=x86-64-v3) vs -O2 -fno-fast-math -march=x86-64 -mno-fma; oracle/flip_evidence.py",
        "stations": n_st, "seconds_each": nb * BS / 1.024e6, "station_blocks": tot_blocks, "flipped_lmr_phase_estimates": tot_flips,
        "flips_per_station_second": tot_flips / (n_st * nb * BS / 1.024e6), "flip_threshold_rad": "half of 0.1 pi / n_est",
        "lmr_worst_block_rms": max(r["lmr_rms_worst_block"] for r in rows),
        "rds_soft_symbols": {"stations_compared": sum(1 for r in rows if r["rds_soft_symbols"]),
                             "symbols": sum(r["rds_soft_symbols"]["symbols"] for r in rows if r["rds_soft_symbols"]),
                             "symbols_over_1e-2": sum(r["rds_soft_symbols"]["symbols_over_1e-2"] for r in rows if r["rds_soft_symbols"]),
                             "worst_station_rms": max((r["rds_soft_symbols"]["rms"] for r in rows if r["rds_soft_symbols"]), default=None),
                             "median_station_rms": float(np.median([r["rds_soft_symbols"]["rms"] for r in rows if r["rds_soft_symbols"]])) if any(r["rds_soft_symbols"] for r in rows) else None,
                             "worst_station_rms_without_those": max((r["rds_soft_symbols"]["rms_without_those"] for r in rows if r["rds_soft_symbols"]), default=None)},
        "per_station": rows}, indent=1))


if __name__ == "__main__":
    main()
