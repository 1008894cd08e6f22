#!/usr/bin/env python3
"""Development: static instruction mix of one kernel of fmd_kernels.hip per barrier-separated phase (hipcc -S with the Makefile's flags).
usage: tools/isa_mix.py <mangled-name-substring> [asm file]   (counts are static: loops and branches not weighted)"""
import re, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
asm = sys.argv[2] if len(sys.argv) > 2 else "/tmp/fmd_kernels.s"
if len(sys.argv) <= 2:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", f"-I{ROOT / 'include'}",
                    "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S", str(ROOT / "fm-radio_amd/csrc/fmd_kernels.hip"), "-o", asm], check=True, capture_output=True)
L = open(asm).read().split("\n")
i0 = next(i for i, l in enumerate(L) if re.match(r"^_Z\w*" + re.escape(sys.argv[1]) + r"\w*:", l))
i1 = next(i for i in range(i0, len(L)) if L[i].strip().startswith("s_endpgm"))
sec, cnt = 0, {}
SLOW = ("v_sin_f32", "v_cos_f32", "v_rcp_f32", "v_perm_b32", "v_fract_f32", "v_rndne_f32", "v_cvt_pk_bf16_f32")
for l in L[i0:i1]:
    t = l.strip().split()
    if not t or t[0].startswith((".", ";", "//")) or t[0].endswith(":"):
        continue
    op = t[0]
    if op == "s_barrier":
        sec += 1
        continue
    kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
            else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    c = cnt.setdefault(sec, {})
    c[kind] = c.get(kind, 0) + 1
    if op in SLOW:
        c[op] = c.get(op, 0) + 1
for s in sorted(cnt):
    print("phase", s, cnt[s])
print("total", {k: sum(c.get(k, 0) for c in cnt.values()) for k in ("valu", "mfma", "salu", "lds", "vmem")})
