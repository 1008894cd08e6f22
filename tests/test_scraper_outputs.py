"""Scraper-compatible outputs: the host-side WAV / RDS-bytes writers against the files the reference's own
`fm_demod_scraper` produces (oracle/_ref/fm_demod_scraper, built from /root/reference).  CPU only: the writers are fed the
oracle's audio, which is bit-identical to the reference's."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oraclelib as O
import synth

ROOT = Path(__file__).resolve().parent.parent
REF_SCRAPER = ROOT / "oracle" / "_ref" / "fm_demod_scraper"


@pytest.mark.skipif(not REF_SCRAPER.exists(), reason="oracle/_ref not built")
def test_wav_and_rds_files_match_reference_scraper(tmp_path):
    bs, nb = 65536, 30
    cap = synth.to_u8(synth.fm_capture(bs * nb, seed=808)["iq"])
    cap.tofile(tmp_path / "cap.u8")
    ref_dir = tmp_path / "ref"
    subprocess.run([str(REF_SCRAPER), "-i", str(tmp_path / "cap.u8"), "-o", str(ref_dir), "-b", str(bs)], check=True, stderr=subprocess.DEVNULL)
    ref_wav = next(ref_dir.glob("*_audio.wav")).read_bytes()
    ref_rds = next(ref_dir.glob("*_rds.bin")).read_bytes()

    o = O.run_chain(cap, bs, 1_024_000, u8=True, streams=["audio", "rds_sym"])
    o["audio"].tofile(tmp_path / "audio.f32")
    o["rds_bytes"].tofile(tmp_path / "rds_bytes.u8")
    exe = tmp_path / "scraper_writer_main"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT / 'fm-radio_amd' / 'host'}", str(ROOT / "tests" / "cpp" / "scraper_writer_main.cpp"), "-o", str(exe)], check=True)
    subprocess.run([str(exe), str(tmp_path / "audio.f32"), str(tmp_path / "rds_bytes.u8"), str(bs // 32), str(tmp_path / "out.wav"), str(tmp_path / "out_rds.bin")], check=True)
    assert (tmp_path / "out.wav").read_bytes() == ref_wav
    assert (tmp_path / "out_rds.bin").read_bytes() == ref_rds
    assert len(ref_rds) > 0 and len(ref_wav) == 44 + nb * (bs // 32) * 4
