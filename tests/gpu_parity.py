"""Shared helper for the GPU parity tests and smoke(): run the HIP library and the CPU oracle on the same
captures, with the oracle given the library's own coefficients, and compare every stream."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oraclelib as O

STREAMS = ["fm_out_iq", "pll_dt", "lpr", "lmr", "rds", "rds_raw_sym", "audio", "lmr_phase"]


def lib_coeffs_to_oracle(k) -> O.Coeffs:
    """fmd_coeffs and fmo_coeffs share one layout (include/fmdemod.h)."""
    out = O.Coeffs()
    assert C.sizeof(out) == C.sizeof(k)
    C.memmove(C.byref(out), C.byref(k), C.sizeof(k))
    return out


def oracle_controls(ctl) -> O.Controls:
    out = O.Controls()
    C.memmove(C.byref(out), C.byref(ctl), C.sizeof(ctl))
    return out


def run_gpu(pkg, caps: np.ndarray, block_size: int, fs: int, use_torch: bool = False, controls=None, per_channel_controls=None,
            pll_kernel: str = "auto", fast_math: bool = False, split_front: bool = False, pll_k16_max=None, pipelined: bool = True):
    """caps: [C, n, 2] float32 or uint8.  Returns dict of per-block concatenated streams [C, ...]."""
    n_ch = caps.shape[0]
    nb = caps.shape[1] // block_size
    dm = pkg.BatchDemod(n_ch, block_size, fs, keep_taps=True, pll_kernel=pll_kernel, fast_math=fast_math, pipelined=pipelined)
    if pll_k16_max is not None:       # (include/fmdemod_debug.h: the pilot-PLL kernel by what is out of lock, from these batch sizes on)
        dm.pll_adaptive(*pll_k16_max) if isinstance(pll_k16_max, tuple) else dm.pll_adaptive(pll_k16_max)
    if split_front:       # (include/fmdemod_debug.h: the first decimator and the front end as two kernels)
        assert dm.L.fmd_debug_split_front(dm.h, 1) == 0
    if controls is not None:
        dm.set_controls(controls)
    if per_channel_controls:
        for ch, ctl in per_channel_controls.items():
            dm.set_controls(ctl, ch)
    names = STREAMS + (["pll_poly"] if fast_math else [])      # tolerance mode: the pilot PLL's span polynomials (always materialised)
    out = {k: [] for k in names + ["rds_sym", "rds_count", "rds_bytes"]}
    rds_bytes = [b"" for _ in range(n_ch)]
    for b in range(nb):
        blk = np.ascontiguousarray(caps[:, b * block_size:(b + 1) * block_size])
        if use_torch:
            import torch
            t = torch.from_numpy(blk).cuda()
            assert dm.process(t) == 0
        else:
            assert dm.process(blk) == 0
        for k in names:
            out[k].append(dm.audio().reshape(n_ch, -1) if k == "audio" else dm.stream(k))
        syms, counts = dm.rds_symbols()
        out["rds_sym"].append([syms[c, :counts[c]].copy() for c in range(n_ch)])
        out["rds_count"].append(counts.copy())
        by, bc = dm.rds_bytes()
        for c in range(n_ch):
            rds_bytes[c] += by[c, :bc[c]].tobytes()
        # raw symbols: only the first counts[c] entries of each row are valid
        raw = out["rds_raw_sym"][-1].reshape(n_ch, -1, 2)
        out["rds_raw_sym"][-1] = [raw[c, :counts[c]].reshape(-1).copy() for c in range(n_ch)]
    res = {}
    for k in names:
        if k == "rds_raw_sym":
            res[k] = [np.concatenate([blk[c] for blk in out[k]]) for c in range(n_ch)]
        else:
            res[k] = np.concatenate(out[k], axis=1)
    res["rds_sym"] = [np.concatenate([blk[c] for blk in out["rds_sym"]]) for c in range(n_ch)]
    res["rds_count"] = np.stack(out["rds_count"], axis=1)
    res["rds_bytes"] = [np.frombuffer(b, dtype=np.uint8) for b in rds_bytes]
    res["coeffs"] = [dm.get_coeffs(c) for c in range(n_ch)]
    dm.close()
    return res


def compare_with_oracle(pkg, caps: np.ndarray, block_size: int, fs: int, use_torch: bool = False, controls=None,
                        per_channel_controls=None, pll_kernel: str = "auto", pll_k16_max=None, pipelined: bool = True) -> dict:
    n_ch = caps.shape[0]
    u8 = caps.dtype == np.uint8
    g = run_gpu(pkg, caps, block_size, fs, use_torch, controls, per_channel_controls, pll_kernel, pll_k16_max=pll_k16_max, pipelined=pipelined)
    report = {"bit_exact": {}, "max_abs": {}, "audio_rms_err": 0.0, "rds_sym_equal_counts": True, "rds_bytes_equal": True}
    for c in range(n_ch):
        ctl = (per_channel_controls or {}).get(c, controls)
        o = O.run_chain(caps[c], block_size, fs, u8=u8, controls=oracle_controls(ctl) if ctl is not None else None,
                        coeffs=lib_coeffs_to_oracle(g["coeffs"][c]),
                        streams=["fm_out_iq", "pll_dt", "lpr", "lmr", "rds", "rds_raw_sym", "audio", "lmr_phase", "rds_sym"])
        for k in STREAMS:
            a = g[k][c] if isinstance(g[k], list) else g[k][c]
            a = np.asarray(a, np.float32).reshape(-1)
            b = o[k].reshape(-1)
            if k == "lmr_phase":
                a = a[-1:]; b = b[-1:]
            same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
            report["bit_exact"][(c, k)] = bool(same)
            if a.shape == b.shape and a.size:
                report["max_abs"][(c, k)] = float(np.max(np.abs(a - b)))
        a, b = g["audio"][c].reshape(-1), o["audio"].reshape(-1)
        report["audio_rms_err"] = max(report["audio_rms_err"], float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2))))
        if not np.array_equal(g["rds_count"][c], o["rds_count"]):
            report["rds_sym_equal_counts"] = False
        if not np.array_equal(g["rds_bytes"][c], o["rds_bytes"]):
            report["rds_bytes_equal"] = False
        report["bit_exact"][(c, "rds_sym")] = bool(g["rds_sym"][c].shape == o["rds_sym"].shape and
                                                   np.array_equal(g["rds_sym"][c].view(np.uint32), o["rds_sym"].view(np.uint32)))
    report["all_bit_exact"] = all(report["bit_exact"].values())
    return report
