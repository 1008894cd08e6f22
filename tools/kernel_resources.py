#!/usr/bin/env python3
"""Regenerate profiles/round<N>/kernel_resources.md (N = argv[1], default 4): the register / LDS / scratch usage of every kernel of fmd_kernels.hip as hipcc reports
it (-Rpass-analysis=kernel-resource-usage, the Makefile's flags).  Runs on the build container (no GPU needed)."""
import re, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
src = ROOT / "fm-radio_amd" / "csrc" / "fmd_kernels.hip"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form",
       f"-I{ROOT / 'include'}", "--cuda-device-only", "-c", str(src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = {}, None
for line in err.splitlines():
    m = re.search(r"remark: (?:    )?([A-Za-z \[\]/]+): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        name = name.split("(")[0].replace("void ", "").replace("fmd::", "")
        name = name.replace("HIP_vector_type<float, 2u>", "cf32").replace("HIP_vector_type<unsigned char, 2u>", "u8")
        cur = rows.setdefault(name, {})
    elif cur is not None:
        cur[k] = v
out = ["# Kernel resources (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage of fm-radio_amd/csrc/fmd_kernels.hip, tools/kernel_resources.py)", "",
       "Occupancy column: wavefronts per SIMD the register allocation allows (512 VGPRs + AGPRs per lane and SIMD); LDS (160 KB per CU) and the",
       "launch (four wavefronts per workgroup for k_pll_sparse and k_rds_sync3) bound it further: k_extract_bp 4 workgroups per CU (40.7 KB of LDS each), k_front_mfma 6-8.", "",
       "| kernel | VGPRs | AGPRs | SGPRs | static LDS bytes / workgroup | scratch B/lane | waves/SIMD by registers |", "|---|---|---|---|---|---|---|"]
for name in sorted(rows):
    r = rows[name]
    out.append(f"| `{name}` | {r.get('VGPRs', '?')} | {r.get('AGPRs', '?')} | {r.get('TotalSGPRs', r.get('SGPRs', '?'))} | {r.get('LDS Size [bytes/block]', '?')} | "
               f"{r.get('ScratchSize [bytes/lane]', '?')} | {r.get('Occupancy [waves/SIMD]', '?')} |")
(ROOT / "profiles" / f"round{sys.argv[1] if len(sys.argv) > 1 else 4}" / "kernel_resources.md").write_text("\n".join(out) + "\n")
print(len(rows), "kernels")
