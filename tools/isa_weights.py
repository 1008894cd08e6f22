#!/usr/bin/env python3
"""Development tool: static VALU mix of the tolerance-mode kernels weighted by the issue cost tools/pk_probe.hip measures on MI355X
(ns per wave-instruction and SIMD at 8 waves/SIMD): v_fma / v_fmac / v_mul / v_add / v_sub f32, v_and, v_mov ~1.0; packed f32 ~2.0;
v_rcp / v_sin / v_cos ~3.4; everything else (compares, selects, shifts, v_perm, v_bfi, v_min / v_max, conversions, DPP moves) ~1.75.
usage: tools/isa_weights.py [kernel-name-substring ...]"""
import collections
import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_mov_b32", "v_or_b32", "v_xor_b32",
        "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
TRANS = {"v_rcp_f32", "v_sin_f32", "v_cos_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}


def weight(op: str, line: str) -> float:
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_mfma"):
        return 0.0
    if base.startswith("v_pk_"):
        return 2.0
    if base in TRANS:
        return 3.4
    if base in FAST and "row_" not in line and "quad_perm" not in line and "wave_sh" not in line:
        return 1.0
    return 1.75


def main() -> None:
    pats = sys.argv[1:] or ["k_front_mfmaI15HIP_vector_typeIfLj2EELi1024ELi0", "k_extract_bp", "k_pll_sparse"]
    with tempfile.TemporaryDirectory() as d:
        s = pathlib.Path(d) / "k.s"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form", f"-I{ROOT / 'include'}", "-S", "--cuda-device-only",
                        str(ROOT / "fm-radio_amd/csrc/fmd_kernels.hip"), "-o", str(s)], check=True, stderr=subprocess.DEVNULL)
        text = s.read_text().splitlines()
    for pat in pats:
        on = False
        ops = collections.Counter(); cost = collections.Counter()
        for ln in text:
            if not on and ln.startswith("_ZN3fmd") and pat in ln and ln.rstrip().endswith(tuple([":"])) is False and ":" in ln:
                on = True
                continue
            if on:
                t = ln.strip()
                if t.startswith("s_endpgm"):
                    break
                m = re.match(r"(v_[a-z0-9_]+)", t)
                if m:
                    ops[m.group(1)] += 1
                    cost[m.group(1)] += weight(m.group(1), t)
        tot = sum(cost.values())
        print(f"{pat}: {sum(ops.values())} VALU instructions, weighted {tot:.0f}")
        for op, c in cost.most_common(14):
            print(f"    {op:28s} x{ops[op]:4d}  {c:7.1f}  {100 * c / tot:5.1f} %")


main()
