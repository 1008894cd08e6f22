#!/usr/bin/env python3
"""Digest of a rocprofv3 --kernel-trace of the pipelined demodulator (development / evidence tool).

    tools/trace_digest.py <dir with *kernel_trace.csv> [steps] [label]

Over the last `steps` launches of every fmd:: kernel (= bench.py's timed region): per stage the launch duration, the
start-to-start interval (= the pipeline's step as that stage sees it) and the idle gap between consecutive launches of the
stage (its stream's idle time: waiting for a producer, a slot, or the dispatcher); over the whole window how long 0, 1, 2, ...
stages ran at once.  What `ms_per_step - max(kernel)` consists of can be read off the gap columns.  Prints one JSON object.
"""
import collections
import csv
import glob
import json
import statistics as st
import sys


def short(name: str) -> str:
    n = name.split("(")[0].replace("void ", "").replace("fmd::", "")
    return n.split("<")[0]


def main() -> None:
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    files = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
    if not files:
        raise SystemExit("no kernel_trace.csv under " + d)
    rows = [r for r in csv.DictReader(open(files[-1])) if "fmd::" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")))
    out = {"label": label, "trace": files[-1].split("/")[-1], "steps": steps, "stages": {}}
    lo, hi = None, None
    win = {}
    for k, v in by.items():
        if len(v) < max(2, steps // 2):
            continue                      # one-off kernels (k_reset, control uploads)
        v.sort()
        v = v[-steps:]
        win[k] = v
        lo = v[0][0] if lo is None else max(lo, v[0][0])     # window in which every stage is in its timed launches
        hi = v[-1][1] if hi is None else min(hi, v[-1][1])
    for k, v in win.items():
        dur = [e - s for s, e, _ in v]
        iv = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
        gap = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        out["stages"][k] = {
            "launches": len(v), "queues": sorted({q for _, _, q in v}),
            "dur_us": {"mean": round(st.mean(dur) / 1e3, 1), "median": round(st.median(dur) / 1e3, 1), "p10": round(sorted(dur)[len(dur) // 10] / 1e3, 1), "p90": round(sorted(dur)[9 * len(dur) // 10] / 1e3, 1)},
            "start_to_start_us": round(st.mean(iv) / 1e3, 1) if iv else None,
            "idle_gap_us": {"mean": round(st.mean(gap) / 1e3, 1), "median": round(st.median(gap) / 1e3, 1), "min": round(min(gap) / 1e3, 1)} if gap else None,
            "busy_frac": round(sum(dur) / max(v[-1][1] - v[0][0], 1), 3),
        }
    # concurrency histogram over [lo, hi]
    ev = []
    for k, v in win.items():
        for s, e, _ in v:
            s2, e2 = max(s, lo), min(e, hi)
            if e2 > s2:
                ev.append((s2, 1)); ev.append((e2, -1))
    ev.sort()
    hist = collections.Counter()
    cur, last = 0, lo
    for t, dlt in ev:
        hist[cur] += t - last
        cur += dlt; last = t
    tot = max(hi - lo, 1)
    out["window_ms"] = round(tot / 1e6, 3)
    out["concurrent_stage_time_frac"] = {str(n): round(hist[n] / tot, 3) for n in sorted(hist)}
    step = [s["start_to_start_us"] for s in out["stages"].values() if s["start_to_start_us"]]
    out["step_us"] = round(st.mean(step), 1) if step else None
    out["longest_launch_us"] = max(s["dur_us"]["mean"] for s in out["stages"].values())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
