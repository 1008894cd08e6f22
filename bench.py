#!/usr/bin/env python3
"""Throughput benchmark of the MI355X FM demodulation hot path (driver contract: see the round prompt).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    (N>1: either launched under torch.distributed.run by the caller, or — when WORLD_SIZE is not set — bench.py starts
     `python -m torch.distributed.run --nproc-per-node N` itself as a CHILD process, before anything touches the GPU,
     relays rank 0's JSON line and exits with the child's return code)

One "step" = one `fmd_process_cf32_dev` call = one block of every channel through the whole chain
(decimating FIRs -> discriminator -> Hilbert -> pilot PLL -> x2/x3 mixers -> audio/RDS decimators -> stereo mix,
RDS AGC + BPSK symbol sync + Manchester).  Workload at N=1: BASELINE.json configs[2] — 4096 synthetic FM channels
@ 256 kSa/s batched on one MI355X, 16384-sample blocks (64 ms), inputs resident in HBM before the timed region.
Channels shard across ranks (weak scaling: 4096 per GPU); the only collective is the per-step audio gather
(RCCL), overlapped with the next step's compute.

The JSON line carries `roofline` (dominant kernel, HIP-event timed inside the library on the processing stream)
and, on rank 0 at N=1, `cpu_baseline` (the CPU oracle = port of the reference, all host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before HIP initialises: one hardware queue per pipeline stream

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "tests", ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3


def algorithmic_flops_per_sample(fs: int) -> float:
    """~385 flop per 256 kSa/s sample for the chain behind the first decimator; that decimator (M x 64 real taps on
    complex samples, one output per M inputs) adds 4 * 64 flop per baseband sample at 1.024 / 2.048 MSa/s."""
    m = fs // 256_000
    return 385.0 / m + (256.0 if m > 1 else 0.0)


FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CUs x 256 flop/clk x 2.4 GHz


def algorithmic_bytes_per_sample(fs: int, u8: bool) -> float:
    """cf32 (or u8) IQ in + stereo f32 audio out + RDS symbol floats out, per baseband sample."""
    in_b = 2.0 if u8 else 8.0
    audio = 8.0 * 32000.0 / fs
    syms = 4.0 * 2375.0 / fs
    return in_b + audio + syms


SIDE_QUEUE_KERNELS = ("k_rds_sync", "k_pll_sparse", "k_pll_span", "k_lmr_phase")
HBM_RIDGE_FLOP_PER_BYTE = 157.3e12 / 8.0e12     # fp32 vector peak over the HBM peak: above it a chain's algorithmic work is compute-bound


def kernel_source_stamp() -> str:
    """sha256 (first 16 hex digits) over the library's kernel and host sources: what ties a row of the committed PMC tables
    (profiles/hbm_traffic.json `_meta.kernel_source_stamp`, written by tools/digest_round.py --install) to the build it was measured on.
    (A git hash would do in the build container; the GPU box's snapshot has no .git.)"""
    import hashlib
    h = hashlib.sha256()
    d = ROOT / "fm-radio_amd" / "csrc"
    for f in sorted(list(d.glob("*.hip")) + list(d.glob("*.inc")) + list(d.glob("*.h")) + list(d.glob("*.cpp"))):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def traffic_meta() -> dict:
    try:
        return json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text()).get("_meta", {})
    except Exception:
        return {}


MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 on the matrix cores (the headline figure with 2:1 sparsity is not priced against)


def roof_sides(fs: int, u8: bool, samples: float, seconds: float) -> dict:
    """Both roofs on ALGORITHMIC work done in `seconds`: HBM bytes against 8 TB/s, and the chain's flops.
    The flops are almost all FIR taps; the tolerance mode runs every FIR on the matrix cores as three bf16 products per fp32 product, so the
    compute roof that applies to them is the bf16 MFMA peak / 3 (833 TFLOP/s of fp32-equivalent work, ridge 104 flop/B), not the fp32
    VECTOR peak (157.3 TFLOP/s, ridge 19.7 flop/B) — which configs[2]'s 42.6 flop/B exceeds and which the measured step
    therefore exceeds too on algorithmic flops (`fp32_vector.frac` > 1 at 1.024 MSa/s).  `bound` names the side the intensity puts the
    configuration under on the pipe its flops run on; all three fractions are printed."""
    bps, fps = algorithmic_bytes_per_sample(fs, u8), algorithmic_flops_per_sample(fs)
    hbm = {"achieved": bps * samples / seconds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    vec = {"achieved": fps * samples / seconds / 1e12, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s"}
    mfma = {"achieved": fps * samples / seconds / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s (fp32 work as 3 bf16 products)"}
    for o in (hbm, vec, mfma):
        o["frac"] = o["achieved"] / o["peak"]
    ridge_mfma = MFMA_BF16_PEAK_TFLOPS / 3.0 * 1e12 / (HBM_PEAK_GBS * 1e9)
    return {"flop_per_byte": fps / bps, "ridge_fp32_vector": HBM_RIDGE_FLOP_PER_BYTE, "ridge_mfma_bf16x3": ridge_mfma,
            "bound": "mfma" if fps / bps > ridge_mfma else "hbm", "hbm": hbm, "fp32_vector": vec, "mfma_bf16x3": mfma}


def lookup_traffic(kernel: str, C: int, fs: int, block: int, u8: bool, fast: bool):
    """HBM bytes per launch of `kernel` in this configuration from the committed PMC table (profiles/hbm_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes, tools/collect_round.sh + tools/digest_round.py), or None if the configuration was not profiled."""
    tf = ROOT / "profiles" / "hbm_traffic.json"
    try:
        tab = json.loads(tf.read_text())
        tail = f"|C={C}|fs={fs}|block={block}|{'u8' if u8 else 'cf32'}|{'fast' if fast else 'exact'}"
        # (the library reports the RDS stage as k_rds_sync whichever of its kernels ran; the trace names the tolerance mode's k_rds_sync3)
        return tab.get(kernel + tail, tab.get(kernel + "3" + tail) if kernel == "k_rds_sync" else None)
    except Exception:
        return None


def chain_traffic(C: int, fs: int, block: int, u8: bool, fast: bool):
    """HBM bytes ONE STEP moves — the sum over the steady chain's kernels of the committed PMC table — or None when one of them was not
    profiled in this configuration.  (The chain of a steady block: the front end (with the first decimator at 1.024 / 2.048 MSa/s), the pilot
    stage, the extract stage, the RDS stage; the exact mode's five kernels.)"""
    if fast:
        names = ["k_front_mfma" if fs == 256_000 else "k_front_pre_mfma", "k_pll_sparse", "k_extract_bp", "k_rds_sync"]
    else:
        names = (["k_predecim"] if fs != 256_000 else []) + ["k_front", "k_pilot_power", "k_pilot_pll", "k_extract", "k_rds_sync"]
    per = {n: lookup_traffic(n, C, fs, block, u8, fast) for n in names}
    if any(v is None for v in per.values()):
        return None, per
    return float(sum(per.values())), per


def dominant_kernel(avg_ms: dict, ms_per_step: float, fast: bool):
    """The kernel `roofline` is quoted on: the longest average launch.  Tolerance mode: its two throughput kernels (k_front_mfma,
    k_extract_bp) take turns on one queue while the serial stages (k_rds_sync: 64 workgroups; the pilot stage rides in the front end's launch) run
    beside them on queues of their own with launches that overlap consecutive blocks'; such a side-queue kernel is the dominant one only
    when its launch is what the step waits for (>= 90 % of the step while the throughput kernels' queue has slack: small batches) —
    otherwise the longest throughput kernel is, and `whole_step_frac` (algorithmic bytes over the whole step) is the figure that says how
    far the step is from the HBM roof.  (A serial launch beside a saturated queue stretches to the length of the step — its wavefronts
    get the issue slots the throughput kernels leave — without being what the step waits for: at 4096 stations the step is 0.256 ms with
    the RDS stage's launch at 0.242 and 0.244 without the stage, profiles/round5/stage_bounds.jsonl.)"""
    if not avg_ms:
        return None, 0.0
    cand = dict(avg_ms)
    if fast:
        main = {k: v for k, v in avg_ms.items() if k not in SIDE_QUEUE_KERNELS}
        side_max = max((v for k, v in avg_ms.items() if k in SIDE_QUEUE_KERNELS), default=0.0)
        if main and (side_max < 0.9 * ms_per_step or sum(main.values()) >= 0.85 * ms_per_step):
            cand = main
    k = max(cand, key=cand.get)
    return k, cand[k]


UNLOCKED_KINDS = ("nopilot", "noise", "zero", "detuned")


def unlocked_plan(n_ch: int, frac: float, kind: str, seed: int) -> np.ndarray:
    """Which channels cannot lock, and how: 0 = normal station, 1 = mono station (no 19 kHz pilot, no L-R, no RDS), 2 = empty
    channel (receiver noise only), 3 = dead channel (all-zero IQ: pilot AGC divides by zero, reference agc.h:12-19), 4 = a station
    whose pilot is 130 Hz off (30 Hz beyond the loop's range: control and integrator on their rails, the loop beats).  The
    affected channels are spread over the batch pseudo-randomly (seeded), as stations are over a band scan."""
    plan = np.zeros(n_ch, np.int8)
    n_bad = int(round(frac * n_ch))
    if n_bad <= 0:
        return plan
    rng = np.random.default_rng(seed)
    idx = rng.choice(n_ch, size=min(n_bad, n_ch), replace=False)
    kinds = {"nopilot": [1], "noise": [2], "zero": [3], "detuned": [4], "mix": [1, 2, 3, 4]}[kind]
    plan[idx] = np.array([kinds[i % len(kinds)] for i in range(idx.size)], np.int8)
    return plan


def synth_block_device(torch, n_ch: int, n_total: int, fs: float, seed: int, device, u8: bool, chunk: int = 256, plan=None):
    """Synthetic multi-channel FM baseband on the GPU: stereo tones + pilot + L-R DSB-SC + BPSK RDS at 57 kHz, FM 75 kHz
    deviation, noise 0.02/rail, scaled x100 (the RTL-SDR u8 amplitude) — the SURVEY §8(d) recipe.  Returns [C, n_total, 2].
    plan: optional per-channel kinds from unlocked_plan()."""
    out = torch.empty((n_ch, n_total, 2), dtype=torch.uint8 if u8 else torch.float32, device=device)
    kinds = torch.zeros(n_ch, dtype=torch.int64, device=device) if plan is None else torch.from_numpy(np.asarray(plan, np.int64)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.arange(n_total, device=device, dtype=torch.float64) / fs
    two_pi = 2.0 * np.pi
    for c0 in range(0, n_ch, chunk):
        c1 = min(n_ch, c0 + chunk)
        m = c1 - c0
        jit = 1.0 + 0.05 * (2.0 * torch.rand((m, 4), generator=g, device=device, dtype=torch.float64) - 1.0)
        ph = two_pi * torch.rand((m, 4), generator=g, device=device, dtype=torch.float64)
        tt = t[None, :]
        left = 0.5 * torch.sin(two_pi * 1000.0 * jit[:, 0:1] * tt + ph[:, 0:1]) + 0.3 * torch.sin(two_pi * 3300.0 * jit[:, 1:2] * tt + ph[:, 1:2])
        right = 0.5 * torch.sin(two_pi * 440.0 * jit[:, 2:3] * tt + ph[:, 2:3]) + 0.3 * torch.sin(two_pi * 5000.0 * jit[:, 3:4] * tt + ph[:, 3:4])
        n_sym = int(np.ceil(n_total / fs * 2375.0)) + 4
        bits = torch.randint(0, 2, (m, n_sym // 2 + 2), generator=g, device=device)
        diff = torch.cumsum(bits, dim=1) % 2
        lvl = 2.0 * diff.to(torch.float64) - 1.0
        sym = torch.stack([lvl, -lvl], dim=2).reshape(m, -1)
        idx = torch.floor(t * 2375.0).to(torch.int64)
        rds = sym[:, idx]
        kd = kinds[c0:c1, None]
        p = two_pi * (19000.0 + 130.0 * (kd == 4).to(torch.float64)) * tt
        stereo = ((kd == 0) | (kd == 4)).to(torch.float64)   # pilot, L-R and RDS present
        carrier = ((kd <= 1) | (kd == 4)).to(torch.float64)  # an FM carrier at all (kind 2: noise only; kind 3: nothing)
        live = (kd != 3).to(torch.float64)
        mpx = 0.40 * (left + right) / 1.6 + stereo * (0.10 * torch.sin(p) + 0.40 * (left - right) / 1.6 * torch.sin(2.0 * p) + 0.06 * rds * torch.sin(3.0 * p))
        phase = two_pi * 75000.0 * torch.cumsum(mpx, dim=1) / fs
        i = carrier * torch.cos(phase) + live * 0.02 * torch.randn((m, n_total), generator=g, device=device, dtype=torch.float64)
        q = carrier * torch.sin(phase) + live * 0.02 * torch.randn((m, n_total), generator=g, device=device, dtype=torch.float64)
        iq = torch.stack([i, q], dim=2)
        if u8:
            out[c0:c1] = torch.clamp(torch.round(127.0 + 100.0 * iq), 0, 255).to(torch.uint8)   # a dead channel is 127 = exactly 0 after the conversion
        else:
            out[c0:c1] = (100.0 * iq).to(torch.float32)
        del left, right, rds, mpx, phase, i, q, iq, sym, lvl, diff, bits
    return out


def synth_wideband_device(torch, centers_hz, n_total: int, fs_in: float, seed: int, device):
    """One wideband capture [n_total, 2] float32 holding a broadcast-FM station (same recipe as synth_block_device, generated
    directly at fs_in) at every centre frequency, summed and scaled so the sum stays within +-100 like an ADC would."""
    two_pi = 2.0 * np.pi
    t = torch.arange(n_total, device=device, dtype=torch.float64) / fs_in
    acc_i = torch.zeros(n_total, device=device, dtype=torch.float64)
    acc_q = torch.zeros(n_total, device=device, dtype=torch.float64)
    for k, fc in enumerate(centers_hz):
        st = synth_block_device(torch, 1, n_total, fs_in, seed + k, device, False, chunk=1)[0].to(torch.float64) / 100.0
        rot = two_pi * ((fc / fs_in * torch.arange(n_total, device=device, dtype=torch.float64)) % 1.0)
        c, s_ = torch.cos(rot), torch.sin(rot)
        acc_i += st[:, 0] * c - st[:, 1] * s_
        acc_q += st[:, 0] * s_ + st[:, 1] * c
        del st, rot, c, s_
    scale = 100.0 / np.sqrt(len(centers_hz)) / 3.0
    del t
    return torch.stack([acc_i * scale, acc_q * scale], dim=1).to(torch.float32).contiguous()


def bench_wideband(args, torch, pkg, device) -> dict:
    """BASELINE configs[4]: one 10 MSa/s capture -> 40 stations by the on-GPU polyphase channeliser -> batched demodulator."""
    fs_in, fs, C = 10_000_000.0, 256_000, 40
    block = 16384
    n_in = block * 625 // 16
    K, W, P = args.steps, args.warmup, max(args.preroll, 0)
    n_res = 8
    centers = (np.arange(C) - (C - 1) / 2.0) * 250e3
    wide = synth_wideband_device(torch, centers, n_res * n_in, fs_in, 1234, device).view(n_res, n_in, 2)
    ch = pkg.Channelizer(fs_in, centers, float(fs), max_input_samples=n_in)
    dm = pkg.BatchDemod(C, block, fs, device=device.index, fast_math=args.fast_math)
    outs = [torch.empty((C, block, 2), dtype=torch.float32, device=device) for _ in range(4)]
    # fmd_process_* would order this stream — and with it the NEXT block's channeliser launch — behind the demodulator's read of the
    # block (two queue hops of ~50 us each around a 165 us kernel).  fmd_submit_* + fmd_wait_input on a side stream instead: the
    # channeliser runs back to back, and a station buffer is rewritten only once the front end has read it (four buffers rotate).
    main, side = torch.cuda.current_stream(device), torch.cuda.Stream(device=device)
    consumed = [None] * len(outs)

    def step(k):
        j = k % len(outs)
        if not args.fast_math:       # (exact mode: its pilot loops hand over from block to block and gain nothing from more blocks in flight)
            dm.process(ch.process(wide[k % n_res], out=outs[j]))
            return
        if consumed[j] is not None:
            main.wait_event(consumed[j])
        y = ch.process(wide[k % n_res], out=outs[j])
        dm.submit(y, ready_stream=main)
        dm.wait_input(side)
        consumed[j] = torch.cuda.Event()
        consumed[j].record(side)

    for k in range(P + W):
        step(k)
    dm.synchronize(); torch.cuda.synchronize(device)
    dm.spec_stats(reset=True)
    dm.profile(0 if args.no_kernel_times else args.kernel_times_mode)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(P + W, P + W + K):
        step(k)
    dm.synchronize(); torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    dm.profile(0)
    ktimes = {k: v for k, v in (dm.profile_read() if not args.no_kernel_times else {}).items() if not k.startswith("gap:")}
    value = C * block * K / el / 1e6
    return {
        "metric": "IQ MSamples/sec demodulated to stereo+RDS per GPU; channels @ real-time",
        "value": value, "unit": "MSa/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": el / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4]: one {fs_in / 1e6:g} MSa/s cf32 capture resident in HBM, {C} FM stations on a 250 kHz raster, "
                               f"on-GPU polyphase channeliser (16/625, 640 taps/phase) -> {fs} Sa/s per station -> full stereo + pilot PLL + RDS",
                   "stations": C, "fs_wideband": fs_in, "fs_baseband": fs, "block_size": block, "preroll_blocks": P,
                   "mode": "fast_math (tolerance)" if args.fast_math else "exact"},
        "wideband_msa_per_s": n_in * K / el / 1e6,
        "realtime_factor": (n_in * K / el) / fs_in,
        "kernels_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in ktimes.items()},
        "speculation": dm.spec_stats(),
    }


def cpu_baseline(fs: int, block: int, budget_s: float = 12.0) -> dict:
    """The CPU oracle (a port of the reference's scalar/AVX path, oracle/fm_oracle.c) on every host core: one
    independent single-channel demodulator per thread (the reference is single-threaded per station), each fed
    its own synthetic capture for `budget_s` seconds of wall time.  Checker code, used here only as the baseline."""
    import oraclelib as O
    import synth
    n_thr = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_blk = 8
    caps = [synth.to_cf32(synth.fm_capture(n_blk * block, fs=float(fs), seed=4000, channel=i)["iq"]) for i in range(min(n_thr, 4))]
    O.lib()
    counts = [0] * n_thr
    stop_at = [0.0]

    def work(i):
        d = O.Demod(block, fs)
        cap = caps[i % len(caps)]
        k = 0
        while time.perf_counter() < stop_at[0]:
            d.process_cf32(cap[(k % n_blk) * block:(k % n_blk + 1) * block])
            k += 1
        counts[i] = k

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
    t0 = time.perf_counter()
    stop_at[0] = t0 + budget_s
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    el = time.perf_counter() - t0
    total = sum(counts) * block
    return {"value": total / el / 1e6, "unit": "MSa/s", "cores": n_thr, "kind": "port",
            "sample": f"{sum(counts)} blocks of {block} cf32 samples @ {fs} Sa/s over {n_thr} threads in {el:.1f} s (oracle/fm_oracle.c, one demodulator per thread)"}


def cpu_reference(budget_s: float = 10.0) -> dict | None:
    """The REAL reference (oracle/_ref/fm_demod_benchmark = reference src/fm_demod_benchmark.cpp built by `make -C oracle ref`)
    at its only rate, 1.024 MSa/s u8: one process per host core, each on its own pass over a page-cached synthetic capture.
    Reported beside cpu_baseline for orientation; None where the prebuilt binary is absent."""
    import subprocess
    import tempfile
    import synth
    exe = ROOT / "oracle" / "_ref" / "fm_demod_benchmark"
    if not exe.exists():
        return None
    n_proc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_proc = max(1, n_proc // 2)  # one per physical core (SMT siblings share the FMA units)
    n = int(1.024e6 * 20) // 65536 * 65536
    cap = synth.to_u8(synth.fm_capture(n, seed=1234)["iq"])
    with tempfile.TemporaryDirectory() as td:
        f = Path(td) / "cap.u8"
        cap.tofile(f)
        subprocess.run([str(exe), "-i", str(f)], stderr=subprocess.DEVNULL, check=True)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([str(exe), "-i", str(f)], stderr=subprocess.DEVNULL) for _ in range(n_proc)]
        for p in procs:
            p.wait()
        el = time.perf_counter() - t0
    return {"value": n * n_proc / el / 1e6, "unit": "MSa/s", "cores": n_proc, "kind": "reference",
            "sample": f"{n_proc} processes x {n} u8 samples @ 1.024 MSa/s (20.5 s of signal each) in {el:.1f} s, reference fm_demod_benchmark incl. RDS decode"}


def host_fed_line(C: int, block: int, fs: int, fast: bool, threads: int = 4, blocks: int = 40) -> dict | None:
    """The PCIe-inclusive rate (never `value`): fm-radio_amd/host/station_ring.hpp — producer threads push u8 IQ into pinned staging
    blocks, H2D copies, the demodulator and the D2H copies of audio + RDS bytes overlap — driven by tests/cpp/station_ring_main.cpp
    (compiled here with g++ against the C ABI and the HIP runtime API).  None when the toolchain or the run fails."""
    import subprocess
    import tempfile
    try:
        with tempfile.TemporaryDirectory() as td:
            exe = Path(td) / "station_ring_main"
            subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}", f"-I{ROOT / 'fm-radio_amd' / 'host'}",
                            str(ROOT / "tests" / "cpp" / "station_ring_main.cpp"), f"-L{ROOT / 'fm-radio_amd' / 'csrc'}", "-lfmdemod", "-L/opt/rocm/lib", "-lamdhip64",
                            "-lpthread", f"-Wl,-rpath,{ROOT / 'fm-radio_amd' / 'csrc'}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True, capture_output=True, timeout=300)
            r = subprocess.run([str(exe), "bench", str(C), str(block), str(fs), str(blocks), str(threads)] + (["fast"] if fast else []),
                               check=True, capture_output=True, text=True, timeout=300)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            d["pcie_bound_msa_per_s"] = 63e9 / 2.0 / 1e6     # 63 GB/s (PCIe Gen5 x16 spec) / 2 B per u8 IQ sample
            d["frac_of_pcie_bound"] = d["host_fed_msa_per_s"] / d["pcie_bound_msa_per_s"]
            d["note"] = "u8 IQ pushed by host threads into pinned staging, H2D + demodulation + D2H of audio and RDS bytes overlapped; PCIe-inclusive, not the headline"
            return d
    except Exception as e:   # noqa: BLE001 - an optional extra line
        return {"error": str(e)[:300]}

def measure_config(torch, pkg, device, label: str, C: int, fs: int, u8: bool, fast: bool, steps: int = 30, preroll: int = 12, warmup: int = 3) -> dict:
    """One extra configuration for the N=1 line's `configs` array: the same timed loop as the headline (resident blocks cycled, loops in
    lock, fmd_submit_*_dev, HIP-event kernel times on every 4th block), shorter.  Returns ms_per_step, MSa/s, the dominant kernel's and the
    whole step's fraction of the HBM roofline on algorithmic bytes."""
    block = fs * 64 // 1000
    if C >= 1024 and fast:
        preroll = max(preroll, 64 if fs == 256_000 else 48)      # (the lead-in also brings the GPU to its steady clocks: ~25 ms of load — profiles/round6/preroll_sweep.txt)
    n_res = 2 if fs > 256_000 else min(8, steps + preroll + warmup)
    x = synth_block_device(torch, C, n_res * block, float(fs), 4321, device, u8)
    x = x.view(C, n_res, block, 2).permute(1, 0, 2, 3).contiguous()
    torch.cuda.synchronize(device)            # (submit() is not ordered behind torch's stream)
    dm = pkg.BatchDemod(C, block, fs, device=device.index, fast_math=fast)
    for k in range(preroll + warmup):
        dm.submit(x[k % n_res])
    dm.synchronize(); torch.cuda.synchronize(device)
    dm.profile(3)
    t0 = time.perf_counter()
    for k in range(preroll + warmup, preroll + warmup + steps):
        dm.submit(x[k % n_res])
    dm.synchronize(); torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    dm.profile(0)
    kt = {k: v[0] / max(v[1], 1) for k, v in dm.profile_read().items() if not k.startswith("gap:")}
    dm.close()
    del x
    torch.cuda.empty_cache()
    bps = algorithmic_bytes_per_sample(fs, u8)
    value = C * block * steps / el / 1e6
    dom = dominant_kernel(kt, el / steps * 1e3, fast)
    return {"config": label, "channels": C, "fs_baseband": fs, "block_size": block, "ingest": "u8" if u8 else "cf32",
            "mode": MODE_TEXT[fast], "steps": steps, "ms_per_step": el / steps * 1e3, "value": value, "unit": "MSa/s",
            "channels_at_realtime": value * 1e6 / fs, "algorithmic_bytes_per_sample": bps,
            "roofline": {"bound": roof_sides(fs, u8, 1.0, 1.0)["bound"], "flop_per_byte": roof_sides(fs, u8, 1.0, 1.0)["flop_per_byte"],
                         "kernel": dom[0], "avg_launch_ms": dom[1], "frac": (bps * C * block / (dom[1] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom[1] else None,
                         "frac_fp32_vector": (algorithmic_flops_per_sample(fs) * C * block / (dom[1] * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS) if dom[1] else None,
                         "whole_step_frac": bps * value * 1e6 / 1e9 / HBM_PEAK_GBS,
                         "whole_step_frac_fp32_vector": algorithmic_flops_per_sample(fs) * value * 1e6 / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                         "traffic": chain_traffic(C, fs, block, u8, fast)[0],       # the chain's HBM bytes per step (committed PMC table), None if not profiled
                         "traffic_ratio": (lambda t: None if t is None else t / (bps * C * block))(chain_traffic(C, fs, block, u8, fast)[0]),
                         "kernels_ms_per_step": kt}}


# what the two arithmetic modes promise and what tests/ assert (tests/test_gpu_fast.py, tests/test_gpu_long.py)
MODE_TEXT = {
    True: "fast_math (tolerance mode: every block of audio / L-R within 1e-4 RMS of the CPU oracle except the blocks behind a flipped L-R phase "
          "estimate of the reference's tracker, bounded by the measured offset difference; whole-run RMS <= 1e-4 asserted on 64 stations x 30 s; "
          "RDS bits identical once the synchroniser is in lock, up to a <= 24-bit shift from acquisition)",
    False: "exact (every output bit-identical to the CPU oracle)",
}
DTYPE_TEXT = {
    True: "f32 (tolerance mode: FIR operands split into 2 x bf16 on the matrix cores, 3 products per tap, fp32 accumulate; everything else fp32)",
    False: "f32",
}


def launch_ranks(n: int) -> int:
    """Start `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child process, pass its
    stderr through, print exactly one JSON line (rank 0's) on stdout and return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preroll", type=int, default=128, help="blocks demodulated before the warmup, untimed like it: the pilot / RDS loops acquire lock "
                    "(a receiver's steady state; 16 blocks = 1 s of signal) and the GPU reaches its steady clocks — from idle the shader clock "
                    "is ~1.6 GHz over the first 5 ms and 2.0-2.2 GHz from ~25 ms of load on, so a 20-step timed region right behind a "
                    "16-block lead-in measured the ramp (258 GSa/s against 273-280 behind 100-200 blocks and 291 sustained over "
                    "12000 steps: profiles/round6/preroll_sweep.txt)")
    ap.add_argument("--channels", type=int, default=4096, help="channels PER GPU (BASELINE configs[2]: 4096)")
    ap.add_argument("--fs", type=int, default=256000, choices=[256000, 1024000, 2048000])
    ap.add_argument("--block", type=int, default=0, help="baseband samples per channel per step (default: 64 ms)")
    ap.add_argument("--u8", action="store_true", help="u8 IQ ingest (2 B/sample) instead of cf32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config3", action="store_true", help="BASELINE configs[3] literally: 8192 stations per GPU (65 536 on 8 GPUs); the default for "
                    "every N is configs[2]'s 4096 per GPU, so that the per-N values the driver compares are one workload (weak scaling)")
    ap.add_argument("--gather", default="rotate", choices=["rotate", "root", "all", "none"],
                    help="N>1: per-step output collective — 'rotate' (default): block k is gathered on rank (k mod N), point-to-point over the "
                         "direct xGMI links, every GPU's PCIe link takes 1/N of the hand-over; 'root': every rank sends its block to rank 0; "
                         "'all': all-gather to every rank; 'none': no collective.  The mode is recorded in the result's `gather` object and in "
                         "`config.gather_mode`: per-link figures of 'rotate' and 'root' runs are not comparable")
    ap.add_argument("--gather-format", default="pcm16", choices=["pcm16", "f32"],
                    help="payload of the audio collective: the 16-bit PCM frames the reference's scraper writes (default) or raw f32")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--kernel-times-mode", type=int, default=3, choices=[1, 2, 3],
                    help="1: HIP timing events on every kernel of every block (costs ~4 %% of the step); 3 (default): every kernel of "
                         "every 4th block, plus the PLL kernel of the block behind it (for the hand-over gap)")
    ap.add_argument("--no-kernel-times", action="store_true", help="do not attach HIP timing events to the kernels of the timed region (no roofline object; ~2 %% faster)")
    ap.add_argument("--pll-kernel", default="auto", choices=["auto", "time_parallel", "time_parallel8", "low_work"],
                    help="force one of the pilot-PLL kernels (default: chosen by batch size; same results either way)")
    ap.add_argument("--no-pipeline", action="store_true", help="run the stages of a block back to back on one stream")
    ap.add_argument("--wideband", action="store_true", help="BASELINE configs[4] instead of configs[2]: one 10 MSa/s capture, 40 stations "
                    "through the on-GPU channeliser, then the batched demodulator (single GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--share-gpu", action="store_true", help="plumbing test: every rank uses cuda:0")
    ap.add_argument("--unlocked-frac", type=float, default=0.0, help="fraction of the channels that cannot hold pilot lock "
                    "(mono stations, empty channels, dead inputs: see --unlocked-kind), spread over the batch")
    ap.add_argument("--unlocked-kind", default="mix", choices=["mix", *UNLOCKED_KINDS],
                    help="nopilot: mono FM station without pilot/L-R/RDS; noise: no carrier, receiver noise only; zero: all-zero IQ "
                         "(the pilot AGC divides by zero, as in the reference); detuned: pilot at 19130 Hz, beyond the loop's range; "
                         "mix: the four in turn")
    ap.add_argument("--deemphasis", type=int, default=0, metavar="US", help="enable the de-emphasis IIR on every channel with this time constant (50 / 75)")
    ap.add_argument("--exact", action="store_true", help="time the exact mode (every output bit-identical to the CPU oracle) as the primary result.  Default: "
                    "FMD_FLAG_FAST_MATH, the tolerance mode — the parity BASELINE.json's north star asks for (audio / L-R within 1e-4 RMS of "
                    "the reference, RDS bits identical: tests/test_gpu_fast.py); the other mode is then timed briefly as `other_mode`")
    ap.add_argument("--fast-math", action="store_true", help="(default) the tolerance mode as the primary result")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the short run of the other arithmetic mode")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (PCIe-inclusive) extra line")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` array (short runs of the other BASELINE configurations)")
    args = ap.parse_args()
    if args.exact and args.fast_math:
        raise SystemExit("bench.py: --exact and --fast-math exclude each other")
    args.fast_math = not args.exact

    # --gpus N without a launcher: start the ranks ourselves, as a CHILD process, before this process has touched the GPU
    # (no torch.cuda / HIP call has happened yet; a process that has initialised the GPU must never exec another program)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with --nproc-per-node equal to --gpus "
                         "(or without a launcher: bench.py starts the ranks itself)")

    import torch
    import torch.distributed as dist

    import fmradio_loader
    pkg = fmradio_loader.load()
    pkg.load_library()  # raises when the HIP extension is missing: there is no fallback to measure

    world = world_env
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    if args.wideband:
        if world != 1:
            raise SystemExit("--wideband is a single-GPU workload")
        print(json.dumps(bench_wideband(args, torch, pkg, device)))
        return

    fs = args.fs
    block = args.block or (fs * 64 // 1000)
    if args.config3:
        args.channels = 8192
    C = args.channels
    K, W, P = args.steps, args.warmup, max(args.preroll, 0)
    n_blocks_resident = min(K + W + P, 8)  # distinct consecutive blocks kept in HBM, cycled
    plan = unlocked_plan(C, args.unlocked_frac, args.unlocked_kind, 99 + rank)
    x = synth_block_device(torch, C, n_blocks_resident * block, float(fs), 1234 + rank, device, args.u8, plan=plan)
    x = x.view(C, n_blocks_resident, block, 2).permute(1, 0, 2, 3).contiguous()  # [blocks][C][N][2]
    # fmd_submit_* reads a block "now", not behind torch's stream: the synthesis must have finished.  (Without this the first blocks
    # were demodulated from half-written memory; NaNs left in the exact mode's loop state cost it 0.85 -> 1.35 ms for good.)
    torch.cuda.synchronize(device)
    dm = pkg.BatchDemod(C, block, fs, device=local_rank, pipelined=not args.no_pipeline, pll_kernel=args.pll_kernel, fast_math=args.fast_math)
    if args.deemphasis:
        ctl = pkg.default_controls()
        ctl.use_deemphasis, ctl.deemphasis_tus = 1, args.deemphasis
        dm.set_controls(ctl)

    do_gather = world > 1 and not args.no_gather and args.gather != "none"
    dm_n_audio = dm.rates.n_audio
    pcm16 = args.gather_format == "pcm16"
    if do_gather:
        rds_cap = int(dm.rds_bytes_tensors()[0].shape[1])
        gather = pkg.AudioGather(dist, torch, C, dm.rates.n_audio, world, device, mode=args.gather,
                                 dtype=torch.int16 if pcm16 else torch.float32, rds_cap=rds_cap)
        gstream = torch.cuda.Stream(device)   # consumes outputs; the submitting stream never waits on them
        dm.set_output_lag(True)

    def step(k: int):
        # the resident blocks are never rewritten: submitted without ordering the caller's stream behind the demodulator's reads
        # (fmd_submit_cf32_dev; with fmd_process_cf32_dev consecutive front-end launches are two cross-queue hand-overs apart)
        dm.submit(x[k % n_blocks_resident])
        # (the gather takes every block's audio one submission later — fmd_set_output_lag: the newest QUEUED outputs, block k - 1's behind
        #  the submission of block k — so the demodulator keeps the schedule of the free-running case; one gather per step all the same)
        if do_gather and k >= 1:
            with torch.cuda.stream(gstream):
                s = gather.slot(k)
                if pcm16:
                    dm.audio_pcm16_into(gather.stage[s], gstream)       # waits for the block's outputs on gstream, then converts
                else:
                    dm.wait_outputs(gstream)
                    gather.stage[s].copy_(dm.audio_tensor(), non_blocking=True)
                # ... and the block's RDS bytes and their counts: the reference's second observer per station (src/app.cpp:27-34)
                rb, rc_ = dm.rds_bytes_tensors()
                sb, sc = gather.stage_rds_views(s)
                sb.copy_(rb, non_blocking=True); sc.copy_(rc_, non_blocking=True)
                # the output views of this block have been read into the staging buffer once gstream gets here: the library
                # must not reuse them earlier, however far the submitting side runs ahead of the collective
                dm.release_outputs(gstream)
                gather.launch(s, k)

    def drain():
        if do_gather:
            with torch.cuda.stream(gstream):
                gather.drain()
            gstream.synchronize()
        dm.synchronize()

    for k in range(P + W):   # pre-roll (pilot PLL / RDS loops acquire lock) then the W warmup steps; consecutive signal
        step(k)
    drain()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    dm.spec_stats(reset=True)
    dm.profile(0 if args.no_kernel_times else args.kernel_times_mode)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(P + W, P + W + K):
        step(k)
    drain()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    el = time.perf_counter() - t0
    dm.profile(False)
    if world > 1:
        tmax = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    # the collective's result is checked (outside the timed region): every rank's last staged block must be what the
    # collector holds for that rank — a slot reused too early or a desynchronised collective shows up here
    gather_verified = None
    if do_gather:
        s_last = (P + W + K - 1) % gather.depth
        mine = (gather.stage[s_last].to(torch.float64).sum().item(), int(gather.stage[s_last].ne(0).sum().item()),
                gather.stage_rds[s_last].to(torch.float64).sum().item())
        sums = [None] * world
        dist.all_gather_object(sums, mine)
        ok = None
        if gather.holds(s_last):       # the rank that collected the last block (mode "rotate": whichever it fell to) checks it
            parts = gather.out[s_last].chunk(world, dim=0)
            ok = all((pt.to(torch.float64).sum().item(), int(pt.ne(0).sum().item())) == tuple(sm[:2]) for pt, sm in zip(parts, sums))
            ok = ok and torch.equal(parts[rank], gather.stage[s_last]) and sums[0][1] > 0
            ok = ok and all(gather.out_rds[s_last][q].to(torch.float64).sum().item() == sums[q][2] for q in range(world))
            ok = bool(ok and torch.equal(gather.out_rds[s_last][rank], gather.stage_rds[s_last]))
        oks = [None] * world
        dist.all_gather_object(oks, ok)
        gather_verified = next((o for o in oks if o is not None), None)
    ktimes = dm.profile_read()
    gaps = {k[4:]: v[0] / max(v[1], 1) for k, v in ktimes.items() if k.startswith("gap:")}   # stream hand-over between launches
    ktimes = {k: v for k, v in ktimes.items() if not k.startswith("gap:")}
    spec = dm.spec_stats()

    gather_note = "" if not do_gather else ({"root": f", per-step gather of audio ({args.gather_format}) + RDS bytes to rank 0 (RCCL)",
                                             "rotate": f", per-step gather of audio ({args.gather_format}) + RDS bytes to rank k mod {world} (RCCL)",
                                             "all": f", per-step all-gather of audio ({args.gather_format}) + RDS bytes (RCCL)"}[args.gather])
    samples_per_step = C * block * world
    value = samples_per_step * K / el / 1e6
    bps = algorithmic_bytes_per_sample(fs, args.u8)

    # dominant kernel: by average launch duration (HIP events on the kernel's own stream), see dominant_kernel(); not the largest
    # accumulated time — the timing events sample the PLL kernel twice as often as the others
    dom = dominant_kernel({k: v[0] / max(v[1], 1) for k, v in ktimes.items()}, el / K * 1e3, args.fast_math)
    roofline = None
    if dom[0]:
        avg_ms = dom[1]
        algo_bytes = bps * C * block  # per launch: every kernel launch covers one block of all local channels
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic_dom = lookup_traffic(dom[0], C, fs, block, args.u8, args.fast_math)
        traffic, traffic_per_kernel = chain_traffic(C, fs, block, args.u8, args.fast_math)
        all_avg = {k: v[0] / max(v[1], 1) for k, v in ktimes.items()}
        longest = max(all_avg, key=all_avg.get) if all_avg else None
        # second roof: the chain is fp32-VALU work; profiles/valu_instructions.json holds the PMC count of VALU
        # wave-instructions one block costs (tools/collect_profiles.sh), the chip issues 1024 SIMDs x clock / 4 of them per second
        valu = None
        vf = ROOT / "profiles" / "valu_instructions.json"
        if vf.exists() and world == 1:
            try:
                vt = json.loads(vf.read_text()).get(f"valu_total_per_block|C={C}|fs={fs}|block={block}|{'u8' if args.u8 else 'cf32'}|{'fast' if args.fast_math else 'exact'}")
                if vt:
                    clk = spec.get("pll_clock_mhz", 2400.0) * 1e6
                    # Two capacities (DESIGN.md "what bounds the step"): a stream of independent wave64 FMAs from >= 2 wavefronts per
                    # SIMD issues every ~2.75 cycles (tools/valu_mix_probe.hip); a single wavefront every ~5, and the co-resident
                    # kernels of this pipeline slow each other as if every instruction held its SIMD for 4 cycles (the classic
                    # wave64-on-SIMD16 cadence, rocprof's VALUBusy convention) - the fraction of THAT capacity is what saturates.
                    cap = 1024 * clk / 4.0
                    valu = {"wave_instructions_per_step": vt, "issue_capacity_per_s": cap, "issue_frac_of_step": vt / (cap * el / K),
                            "cycles_per_instruction_assumed": 4.0, "frac_at_probe_rate_2p75": vt / (1024 * clk / 2.75 * el / K)}
            except Exception:
                valu = None
        # Which roof: roof_sides() — configs[2] asks 385 flop for 9.04 B per sample = 42.6 flop/B: above the fp32 VECTOR
        # ridge (19.7), below the ridge of the matrix cores its FIRs run on (104): `bound` names the side, `achieved / peak / unit / frac` are that
        # side's for the dominant kernel's launch, every side is printed beside it (and for the whole step in `whole_step`).
        sides = roof_sides(fs, args.u8, C * block, avg_ms * 1e-3)
        step_sides = roof_sides(fs, args.u8, C * block * K, el) if world == 1 else None
        side = sides["mfma_bf16x3" if sides["bound"] == "mfma" else "hbm"]
        meta = traffic_meta()
        stamp = kernel_source_stamp()
        roofline = {"bound": sides["bound"], "kernel": dom[0], "achieved": side["achieved"], "peak": side["peak"], "unit": side["unit"],
                    "frac": side["frac"],
                    # the longest launch of the step whatever its queue (the serial RDS stage runs beside the throughput kernels on a queue of its own)
                    "kernel_longest": None if longest is None else {"kernel": longest, "avg_launch_ms": all_avg[longest],
                                                                    "frac_hbm": algo_bytes / (all_avg[longest] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    "flop_per_byte": sides["flop_per_byte"], "ridge_fp32_vector": sides["ridge_fp32_vector"], "ridge_mfma_bf16x3": sides["ridge_mfma_bf16x3"],
                    "hbm": sides["hbm"], "fp32_vector": sides["fp32_vector"], "mfma_bf16x3": sides["mfma_bf16x3"],
                    "whole_step": None if step_sides is None else {"hbm_frac": step_sides["hbm"]["frac"], "fp32_vector_frac": step_sides["fp32_vector"]["frac"],
                                                                   "mfma_bf16x3_frac": step_sides["mfma_bf16x3"]["frac"]},
                    # HBM bytes one step really moves: the CHAIN's (every kernel of a steady block), beside the algorithmic bytes `achieved` is made of
                    "traffic": traffic, "traffic_ratio": None if traffic is None else traffic / algo_bytes,
                    "traffic_per_kernel": traffic_per_kernel, "traffic_dominant_kernel": traffic_dom,
                    "traffic_source": None if traffic is None else "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "configuration (tools/collect_round.sh), committed; not re-measured by this run",
                    # the table's rows were measured on the library whose sources hash to `traffic_measured_on`; this run's library: `kernel_source_stamp`
                    "traffic_measured_on": meta.get("kernel_source_stamp"), "kernel_source_stamp": stamp,
                    "traffic_stale": None if traffic is None else (meta.get("kernel_source_stamp") != stamp),
                    "avg_launch_ms": avg_ms,
                    "algorithmic_bytes_per_launch": algo_bytes,
                    "whole_step_frac": (bps * C * block * K / el / 1e9) / HBM_PEAK_GBS if world == 1 else None,
                    "kernels_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in ktimes.items()},
                    "handover_ms": gaps,
                    "valu": valu,
                    "flops": None if world != 1 else {"per_sample": algorithmic_flops_per_sample(fs), "achieved_tflops": algorithmic_flops_per_sample(fs) * value / 1e6,
                                                       "peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
                                                       "frac": algorithmic_flops_per_sample(fs) * value / 1e6 / FP32_VECTOR_PEAK_TFLOPS}}

    out = {
        "metric": "IQ MSamples/sec demodulated to stereo+RDS per GPU; channels @ real-time",
        "value": value,
        "unit": "MSa/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": el / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_TEXT[args.fast_math],
        "data": "synthetic",
        "config": {"workload": ("BASELINE configs[2]: " if (C, fs, args.u8) == (4096, 256000, False) else "variant of BASELINE configs[2]: ") +
                               f"{C} synthetic FM channels/GPU @ {fs} Sa/s, {block}-sample blocks, "
                               f"{'u8' if args.u8 else 'cf32'} IQ resident in HBM, full stereo + pilot PLL + RDS" +
                               ("" if world == 1 else f"; {C * world} stations over {world} GPUs" +
                                (" = BASELINE configs[3]" if (C, world) == (8192, 8) else " (BASELINE configs[3] literally, 65536 stations over 8 GPUs, is --config3: 8192 per GPU)")),
                   "channels_per_gpu": C, "fs_baseband": fs, "block_size": block, "ingest": "u8" if args.u8 else "cf32",
                   "preroll_blocks": P, "resident_signal": f"{n_blocks_resident} consecutive blocks cycled (phase-continuous for the 19 kHz pilot)",
                   "mode": MODE_TEXT[args.fast_math],
                   "unlocked_frac": args.unlocked_frac, "unlocked_kind": args.unlocked_kind if args.unlocked_frac > 0 else None,
                   "deemphasis_us": args.deemphasis or None,
                   "parallelism": f"channel-sharded x{world}" + gather_note,
                   "gather_mode": args.gather if do_gather else None},
        "gather_verified": gather_verified,
        # what the per-step gather asks of the collector's xGMI links (one direct link per peer): at throughput-mode rates this, not the
        # demodulation, can bound the multi-GPU step
        "gather": None if not do_gather else (lambda per_rank: {
            "mode": args.gather, "format": args.gather_format, "payload": "audio + RDS byte buffers + counts",
            "bytes_per_rank_per_step": per_rank,
            # root: every block of a rank crosses its link to the one collector; rotate: 1 / world of them cross each of its links
            "gb_per_s_per_link_at_this_rate": per_rank * K / el / 1e9 / (world if args.gather == "rotate" else 1),
            "xgmi_link_gb_per_s_per_direction": 77,
            "collector_ingress_gb_per_s": (world - 1) * per_rank * K / el / 1e9 / (world if args.gather == "rotate" else 1) if args.gather != "all" else None,
        })(C * dm_n_audio * 2 * (2 if pcm16 else 4) + gather.rds_bytes),
        "channels_at_realtime": value * 1e6 / fs,
        "msa_per_gpu": value / world,
        "roofline": roofline,
        # pilot PLL kernel in the timed region: samples committed per 16-sample speculative span, chunks that fell back to the
        # plain serial iteration, spans redone with the reference forms — same results either way
        "speculation": spec,
    }
    dm.close()
    if world == 1 and not args.no_other_mode and not args.no_pipeline:
        # the same workload in the other arithmetic mode, timed the same way over at most 20 steps (no kernel timing events)
        dm2 = pkg.BatchDemod(C, block, fs, device=local_rank, fast_math=not args.fast_math)
        if args.deemphasis:
            dm2.set_controls(ctl)
        K2 = min(K, 20)
        for k in range(P + W):
            dm2.submit(x[k % n_blocks_resident])
        dm2.synchronize(); torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for k in range(P + W, P + W + K2):
            dm2.submit(x[k % n_blocks_resident])
        dm2.synchronize(); torch.cuda.synchronize(device)
        el2 = time.perf_counter() - t0
        dm2.close()
        out["other_mode"] = {"mode": MODE_TEXT[not args.fast_math],
                             "value": C * block * K2 / el2 / 1e6, "unit": "MSa/s", "steps": K2, "ms_per_step": el2 / K2 * 1e3}
    if rank == 0 and world == 1 and not args.no_configs and not args.no_pipeline:
        # the other configurations of BASELINE.json, short runs of the same loop: the reference's native rate and
        # capture format (src/app.cpp:56-65), configs[1] in both modes, the per-GPU shard of configs[3]
        del x
        x = None
        torch.cuda.empty_cache()
        out["configs"] = [
            measure_config(torch, pkg, device, "4096 ch @ 1.024 MSa/s cf32 (the reference's native rate)", 4096, 1_024_000, False, True),
            measure_config(torch, pkg, device, "4096 ch @ 1.024 MSa/s u8 (the reference's capture format)", 4096, 1_024_000, True, True),
            measure_config(torch, pkg, device, "configs[1]: 1 ch @ 2.048 MSa/s, tolerance mode", 1, 2_048_000, False, True, steps=60),
            measure_config(torch, pkg, device, "configs[1]: 1 ch @ 2.048 MSa/s, exact mode", 1, 2_048_000, False, False, steps=60, preroll=24),
            measure_config(torch, pkg, device, "configs[3] per-GPU shard: 8192 ch @ 256 kSa/s", 8192, 256_000, False, True),
            # (exact mode: the pilot loop's speculation runs at full length only once every station is in lock: ~20 blocks)
            measure_config(torch, pkg, device, "configs[2] in the exact mode: 4096 ch @ 256 kSa/s", 4096, 256_000, False, False, steps=20, preroll=24),
        ]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        x = None
        torch.cuda.empty_cache()
        if not args.no_host_fed:
            out["host_fed_u8"] = host_fed_line(C, block, fs, args.fast_math)
        # cpu_baseline: the REAL reference (oracle/_ref/fm_demod_benchmark, prebuilt from the reference's own sources) on the box's host
        # cores, on its native workload — 1.024 MSa/s u8 captures, one single-threaded demodulator per physical core;
        # cpu_port: the oracle's scalar restatement at THIS run's rate and format (`kind: port`), which is what cpu_baseline was through
        # round 5 and still is where the reference binary did not travel
        port = cpu_baseline(fs, block)
        ref = cpu_reference()
        if ref is not None:
            ref["workload_note"] = ("the reference has one rate and one capture format (Fs_baseband = 1 024 000, u8: src/app.cpp:56-65, "
                                    "broadcast_fm_demod.cpp:68); its MSa/s are baseband samples of THAT workload, four per 256 kSa/s sample of configs[2]")
            out["cpu_baseline"] = ref
            out["cpu_port"] = port
        else:
            out["cpu_baseline"] = port
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
