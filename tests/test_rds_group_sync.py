"""Host-side RDS group synchroniser (fm-radio_amd/host/rds_group_sync.hpp) against the reference's own
RDS_Group_Sync + decoder log (oracle/_ref/fm_ref_dump stderr), on a clean and on a noisy capture (block errors, single-bit
corrections, loss of sync and re-lock)."""
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oraclelib as O
import synth

ROOT = Path(__file__).resolve().parent.parent


def _ref_log(cap, tmp):
    cap.tofile(tmp / "cap.u8")
    (tmp / "out").mkdir(exist_ok=True)
    p = subprocess.run([str(O.REF_DUMP), "chain", str(tmp / "cap.u8"), str(tmp / "out"), "65536"], check=True, capture_output=True, text=True)
    groups = re.findall(r"\[rds_decoder\] (\[group\] \[[0-9A-F\- ]+\])", p.stderr)
    locks = [int(v) for v in re.findall(r"Locked onto block A after (\d+) bits", p.stderr)]
    return groups, locks, np.fromfile(tmp / "out" / "rds_bytes.u8", dtype=np.uint8)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("noise,seed", [(0.02, 21), (0.45, 22)])
def test_group_sync_matches_reference(tmp_path, noise, seed):
    cap = synth.to_u8(synth.fm_capture(60 * 65536, seed=seed, noise_sigma=noise)["iq"])
    ref_groups, ref_locks, ref_bytes = _ref_log(cap, tmp_path)
    assert len(ref_groups) > 20
    exe = tmp_path / "group_sync_main"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT / 'fm-radio_amd' / 'host'}", str(ROOT / "tests" / "cpp" / "group_sync_main.cpp"), "-o", str(exe)], check=True)
    ref_bytes.tofile(tmp_path / "rds.bin")
    out = subprocess.run([str(exe), str(tmp_path / "rds.bin")], check=True, capture_output=True, text=True).stdout.splitlines()
    groups = [l for l in out if l.startswith("[group]")]
    locks = [int(l.split()[-2]) for l in out if l.startswith("Locked")]
    assert groups == ref_groups
    assert locks == ref_locks
    if noise > 0.1:
        assert any("----" in g for g in groups), "noisy capture was meant to exercise invalid blocks"
