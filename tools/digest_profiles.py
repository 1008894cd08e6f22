#!/usr/bin/env python3
"""Turn gpurun_out/prof/ (written by tools/collect_profiles.sh on the GPU box) into the tracked summaries under profiles/."""
import csv, glob, json, os, sys, collections, pathlib


def newest(pattern):
    """gpurun merges every run's files into gpurun_out/: take the most recent match."""
    return max(glob.glob(pattern), key=os.path.getmtime)


ROOT = pathlib.Path(__file__).resolve().parent.parent
src = ROOT / "gpurun_out" / "prof"
rnd = sys.argv[1] if len(sys.argv) > 1 else "round2"
dst = ROOT / "profiles" / rnd
dst.mkdir(parents=True, exist_ok=True)

bench = json.loads((src / "bench_default.json").read_text().strip().splitlines()[-1])
# the default command's files keep their names; the exact mode's (BENCH_ARGS=--exact) carry a suffix
sfx = "" if str(bench["config"].get("mode", "")).startswith("fast") else "_exact"
(dst / f"bench_default{sfx}.json").write_text(json.dumps(bench, indent=1) + "\n")
under = json.loads((src / "bench_under_rocprof.json").read_text().strip().splitlines()[-1])

stats = newest(str(src / "stats" / "*" / "*kernel_stats.csv"))
rows = [r for r in csv.DictReader(open(stats))]
with open(dst / f"bench_default_kernel_stats{sfx}.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline   (durations in ns)\n")
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        if len(r["Name"]) > 160: r["Name"] = r["Name"][:157] + "..."   # torch's synthetic-data kernels carry kilobyte-long names
        w.writerow(r)
    # the --stats averages include the pre-roll and warmup launches (loop acquisition); the timed region is the last `steps` launches
    trace = newest(str(src / "stats" / "*" / "*kernel_trace.csv"))
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    K = under["steps"]
    timed = {}
    for name, v in per.items():
        if "fmd::" in name:
            v.sort()
            timed[name.split("(")[0].replace("void ", "")] = round(sum(d for _, d in v[-K:]) / K / 1e6, 4)
    f.write("# average over the LAST %d launches of each kernel (= bench.py's timed region), ms, from the kernel trace: %s\n" % (K, json.dumps(timed)))
    f.write("# bench.py's own HIP-event averages in the same run (ms): " + json.dumps(under["roofline"]["kernels_ms_per_step"]) + "\n")

def pmc(counter):
    f = newest(str(src / f"pmc_{counter}" / "*" / "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
    for (_, name), v in per_dispatch.items():
        agg[name].append(v)
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in agg.items()}   # steady-state half of the launches

fetch, write = pmc("fetch"), pmc("write")
cfg = bench["config"]
mode = "fast" if str(cfg.get("mode", "")).startswith("fast") else "exact"
key_tail = f"|C={cfg['channels_per_gpu']}|fs={cfg['fs_baseband']}|block={cfg['block_size']}|{cfg['ingest']}|{mode}"
names = {"k_front": "k_front", "k_pilot_power": "k_pilot_power", "k_pilot_pll": "k_pilot_pll", "k_pll_fast": "k_pll_fast", "k_extract": "k_extract",
         "k_rds_sync": "k_rds_sync"}
try:
    old_traffic = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())
except Exception:
    old_traffic = {}
traffic, lines = {}, []
for short in names:
    fk = [v for k, v in fetch.items() if short in k]
    wk = [v for k, v in write.items() if short in k]
    if not fk:
        continue
    fr, wr = fk[0], wk[0]
    total = (2.0 * fr + wr) * 1024.0     # KiB; gfx950 FETCH_SIZE counts half of a 16 B/lane stream (MI355X_MICROARCH.md, HBM section)
    traffic[short + key_tail] = total
    lines.append(f"| {short} | {fr:.0f} | {wr:.0f} | {total / 1e6:.1f} MB |")
old_traffic.update(traffic)
(ROOT / "profiles" / "hbm_traffic.json").write_text(json.dumps(old_traffic, indent=1) + "\n")
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
(dst / f"hbm_traffic_pmc{sfx}.md").write_text(
    "# HBM traffic per launch, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)\n\n"
    "Command: `rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 4 --warmup 1 --preroll 16 --no-cpu-baseline --no-other-mode --no-pipeline`"
    " (and WRITE_SIZE), " + cfg["workload"] + ".\nCounter unit KiB; reads doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE = 1/2 of a 16 B/lane coalesced stream)."
    " Averages over the steady-state launches.\n\n| kernel | FETCH_SIZE (KiB, raw) | WRITE_SIZE (KiB) | corrected HBM bytes / launch |\n|---|---|---|---|\n"
    + "\n".join(lines) + f"\n\nSum over the chain: {sum(traffic.values()) / 1e6:.0f} MB per block (algorithmic: {algo / 1e6:.1f} MB).\n")
# VALU instruction counts per launch (wave-instructions) -> how close the pipelined step is to the chip's issue capacity
try:
    f = newest(str(src / "pmc_insts" / "*" / "*counter_collection.csv"))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (_, name), cs in per.items():
        for k, v in cs.items():
            agg[name][k].append(v)
    insts = {}
    for short in names:
        hit = [n for n in agg if short in n]
        if hit:
            v = agg[hit[0]]
            insts[short] = {k: sum(x[len(x) // 2:]) / len(x[len(x) // 2:]) for k, x in v.items()}
    total_valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in insts.values())
    clock_mhz = bench.get("speculation", {}).get("pll_clock_mhz", 2400.0)
    # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles and SIMD: the cadence the co-resident kernels of the pipeline behave
    # by; streams of independent FMAs from >= 2 wavefronts per SIMD reach one per 2.75 (tools/valu_mix_probe.hip)
    cap = 1024 * clock_mhz * 1e6 / 4.0
    step_s = bench["ms_per_step"] * 1e-3
    summary = {"wave_instructions_per_block": insts, "valu_total_per_block": total_valu, "simd_issue_capacity_per_s": cap,
               "valu_issue_fraction_of_step": total_valu / (cap * step_s), "cycles_per_instruction_assumed": 4.0, "clock_mhz_used": clock_mhz, "ms_per_step": bench["ms_per_step"]}
    (dst / f"valu_instructions_pmc{sfx}.json").write_text(json.dumps(summary, indent=1) + "\n")
    try:
        old_v = json.loads((ROOT / "profiles" / "valu_instructions.json").read_text())
    except Exception:
        old_v = {}
    old_v["valu_total_per_block" + key_tail] = total_valu
    (ROOT / "profiles" / "valu_instructions.json").write_text(json.dumps(old_v, indent=1) + "\n")
    print(json.dumps(summary, indent=1))
except Exception as e:   # older gpurun_out without the pass
    print("no instruction-count pass:", e)
print(json.dumps(traffic, indent=1))
print(open(dst / f"bench_default_kernel_stats{sfx}.csv").read()[:3000])
