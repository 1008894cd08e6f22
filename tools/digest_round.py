#!/usr/bin/env python3
"""Reduce the outputs of tools/collect_round.sh to per-kernel averages (one JSON), on the GPU box; `tools/digest_round.py --install [round]`
in the build container then copies the digests into profiles/round<N>/ and refreshes profiles/hbm_traffic.json /
profiles/valu_instructions.json (the committed tables bench.py looks `traffic` and the VALU issue fraction up in)."""
import collections
import csv
import glob
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent


def short(name: str) -> str:
    return name.split("(")[0].replace("void ", "").replace("fmd::", "").split("<")[0]


def pmc_dir(d: pathlib.Path) -> dict:
    out = {}
    for f in glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "fmd::" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], short(r["Kernel_Name"]))][r["Counter_Name"]] += float(r["Counter_Value"])
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for (_, k), cs in per.items():
            for c, v in cs.items():
                agg[k][c].append(v)
        for k, cs in agg.items():
            if k == "k_reset" or max(len(v) for v in cs.values()) < 3:     # (k_pll_span: a station's first 8192 samples only, not part of the steady chain)
                continue
            out.setdefault(k, {}).update({c: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for c, v in cs.items()})   # steady-state half of the launches
    return out


def reduce(o: pathlib.Path, sfx: str) -> None:
    bench = json.loads((o / "bench_default.json").read_text().strip().splitlines()[-1])
    under = json.loads((o / "bench_under_rocprof.json").read_text().strip().splitlines()[-1])
    counters = {}
    for sub in ("fetch", "write", "sq_a", "sq_b", "sq_c"):
        for k, cs in pmc_dir(o / f"pmc_{sub}").items():
            counters.setdefault(k, {}).update(cs)
    stats = sorted(glob.glob(str(o / "stats" / "**" / "*kernel_stats.csv"), recursive=True))
    rows = [r for r in csv.DictReader(open(stats[-1]))] if stats else []
    for r in rows:
        if len(r["Name"]) > 160:
            r["Name"] = r["Name"][:157] + "..."
    trace = sorted(glob.glob(str(o / "stats" / "**" / "*kernel_trace.csv"), recursive=True))
    timed = {}
    if trace:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(trace[-1])):
            if "fmd::" in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        K = under["steps"]
        for k, v in per.items():
            if len(v) >= K:
                v.sort()
                timed[k] = round(sum(d for _, d in v[-K:]) / K / 1e6, 4)
    (o / "digest.json").write_text(json.dumps({
        "bench_default": bench, "bench_under_rocprof": {k: under[k] for k in ("value", "ms_per_step", "steps")} | {"kernels_ms_per_step": under["roofline"]["kernels_ms_per_step"]},
        "kernel_stats_rows": rows, "kernel_trace_avg_ms_last_steps": timed, "counters_per_launch": counters,
        "trace_digest": json.loads((o / "trace_digest.json").read_text() or "{}")}, indent=1))


def install(rnd: str = "5") -> None:
    dst = ROOT / "profiles" / f"round{rnd}"
    dst.mkdir(parents=True, exist_ok=True)
    traffic_tab = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())
    valu_tab = json.loads((ROOT / "profiles" / "valu_instructions.json").read_text())
    for sfx in ("", "_exact", "_u8", "_1024k", "_1024k_u8", "_8192", "_1ch"):
        f = ROOT / "gpurun_out" / f"r{rnd}prof{sfx}" / "digest.json"
        if not f.exists():
            continue
        d = json.loads(f.read_text())
        bench = d["bench_default"]
        cfg = bench["config"]
        mode = "fast" if str(cfg.get("mode", "")).startswith("fast") else "exact"
        tail = f"|C={cfg['channels_per_gpu']}|fs={cfg['fs_baseband']}|block={cfg['block_size']}|{cfg['ingest']}|{mode}"
        (dst / f"bench_default{sfx}.json").write_text(json.dumps(bench, indent=1) + "\n")
        with open(dst / f"bench_default_kernel_stats{sfx}.csv", "w") as fh:
            fh.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-other-mode --no-configs --no-host-fed   (durations in ns)\n")
            if d["kernel_stats_rows"]:
                w = csv.DictWriter(fh, fieldnames=list(d["kernel_stats_rows"][0].keys()))
                w.writeheader()
                w.writerows(d["kernel_stats_rows"])
            fh.write("# average over the LAST %d launches of each kernel (= bench.py's timed region), ms, from the kernel trace: %s\n" % (d["bench_under_rocprof"]["steps"], json.dumps(d["kernel_trace_avg_ms_last_steps"])))
            fh.write("# bench.py's own HIP-event averages in the same run (ms): " + json.dumps(d["bench_under_rocprof"]["kernels_ms_per_step"]) + "\n")
        (dst / f"pipelined_trace_digest{sfx}.json").write_text(json.dumps(d["trace_digest"], indent=1) + "\n")
        ctr = d["counters_per_launch"]
        lines, total, valu_total = [], 0.0, 0.0
        for k, cs in ctr.items():
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                b = (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0     # KiB; gfx950 FETCH_SIZE counts half of a 16 B/lane stream (MI355X_MICROARCH.md, HBM section)
                traffic_tab[k + tail] = b
                total += b
                lines.append(f"| {k} | {cs['FETCH_SIZE']:.0f} | {cs['WRITE_SIZE']:.0f} | {b / 1e6:.1f} MB |")
            valu_total += cs.get("SQ_INSTS_VALU", 0.0)
        algo = bench["roofline"]["algorithmic_bytes_per_launch"]
        (dst / f"hbm_traffic_pmc{sfx}.md").write_text(
            "# HBM traffic per launch, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_round.sh)\n\n" + cfg["workload"] + f", mode: {mode}.\n"
            "Counter unit KiB; reads doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE = 1/2 of a 16 B/lane coalesced stream). Averages over the steady-state launches of the un-pipelined run.\n\n"
            "| kernel | FETCH_SIZE (KiB, raw) | WRITE_SIZE (KiB) | corrected HBM bytes / launch |\n|---|---|---|---|\n" + "\n".join(lines) +
            f"\n\nSum over the chain: {total / 1e6:.0f} MB per block (algorithmic: {algo / 1e6:.1f} MB, ratio {total / algo:.2f}).\n")
        valu_tab["valu_total_per_block" + tail] = valu_total
        keep = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_VALU_MFMA_BUSY_CYCLES",
                "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA",
                "SQ_ACTIVE_INST_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
        (dst / f"sq_counters_pmc{sfx}.json").write_text(json.dumps({
            "what": "per-kernel averages per launch of the un-pipelined bench (tools/collect_round.sh: one rocprofv3 --pmc run per counter set); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* "
                    "count quad-cycles summed over wavefronts, SQ_VALU_MFMA_BUSY_CYCLES cycles, SQ_INSTS_* wave-instructions (MI355X_MICROARCH.md)",
            "workload": cfg["workload"], "mode": mode, "valu_total_per_block": valu_total,
            "kernels": {k: {c: v for c, v in cs.items() if c in keep} for k, cs in ctr.items()}}, indent=1) + "\n")
    # stamp the table with the library it was measured on (bench.py: kernel_source_stamp / traffic_stale)
    import importlib.util, subprocess
    spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py"); bench_mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench_mod)
    try:
        git = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        git = ""
    traffic_tab["_meta"] = {"kernel_source_stamp": bench_mod.kernel_source_stamp(), "git_head_at_install": git, "round": rnd}
    (ROOT / "profiles" / "hbm_traffic.json").write_text(json.dumps(traffic_tab, indent=1) + "\n")
    (ROOT / "profiles" / "valu_instructions.json").write_text(json.dumps(valu_tab, indent=1) + "\n")
    print("installed into", dst)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--install":
        install(sys.argv[2] if len(sys.argv) > 2 else "5")
    else:
        reduce(pathlib.Path(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else "")
