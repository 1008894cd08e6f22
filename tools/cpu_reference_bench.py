#!/usr/bin/env python3
"""Time the REAL reference (oracle/_ref/fm_demod_benchmark, built from /root/reference by `make -C oracle ref`) on the
host cores: one process per core (the reference is single-threaded), each reading its own page-cached copy of a synthetic
u8 capture at the reference's only rate (1.024 MSa/s), as BASELINE.md §3 plans.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np  # noqa: E402
import synth  # noqa: E402

exe = ROOT / "oracle" / "_ref" / "fm_demod_benchmark"
n_proc = int(sys.argv[1]) if len(sys.argv) > 1 else len(os.sched_getaffinity(0))
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
n = int(1.024e6 * seconds) // 65536 * 65536
cap = synth.to_u8(synth.fm_capture(n, seed=1234)["iq"])
with tempfile.TemporaryDirectory() as td:
    f = Path(td) / "cap.u8"
    cap.tofile(f)
    subprocess.run([str(exe), "-i", str(f)], stderr=subprocess.DEVNULL, check=True)  # warm the page cache
    t0 = time.perf_counter()
    procs = [subprocess.Popen([str(exe), "-i", str(f)], stderr=subprocess.DEVNULL) for _ in range(n_proc)]
    for p in procs:
        p.wait()
    el = time.perf_counter() - t0
print(json.dumps({"kind": "reference", "processes": n_proc, "msa_per_s": n * n_proc / el / 1e6, "seconds": el,
                  "samples_per_process": n, "per_process_msa_per_s": n / el / 1e6}))
