"""CPU tests (no GPU): bench.py's bookkeeping around the JSON line — which kernel `roofline` is quoted on, and the chain-traffic lookup."""
import importlib.util
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_the_dominant_kernel_is_a_serial_side_queue_launch_only_where_the_throughput_queue_has_slack():
    b = _bench()
    # 4096 stations, round 5: the RDS launch stretches to 94 % of the step beside a saturated queue (front + extract = 95 % of it): not dominant
    k, ms = b.dominant_kernel({"k_front_mfma": 0.152, "k_extract_bp": 0.089, "k_rds_sync": 0.241}, 0.2549, True)
    assert (k, ms) == ("k_front_mfma", 0.152)
    # 1024 stations: the throughput kernels fill 70 % of the step, the RDS launch 95 %: it is what the step waits for
    k, ms = b.dominant_kernel({"k_front_mfma": 0.058, "k_extract_bp": 0.026, "k_rds_sync": 0.121}, 0.1272, True)
    assert k == "k_rds_sync"
    # a short serial launch is never dominant; the exact mode has no side queues: plainly the longest launch
    assert b.dominant_kernel({"k_front_mfma": 0.15, "k_extract_bp": 0.09, "k_rds_sync": 0.1, "k_pll_sparse": 0.06}, 0.26, True)[0] == "k_front_mfma"
    assert b.dominant_kernel({"k_front": 0.385, "k_pilot_pll": 0.762, "k_extract": 0.72, "k_rds_sync": 0.442}, 0.86, False)[0] == "k_pilot_pll"
    assert b.dominant_kernel({}, 0.3, True) == (None, 0.0)


def test_chain_traffic_comes_from_the_committed_pmc_table_for_the_bench_workload():
    b = _bench()
    total, per = b.chain_traffic(4096, 256_000, 16384, False, True)
    assert total is not None and abs(total - sum(per.values())) < 1.0
    assert {"k_front_mfma", "k_extract_bp"} <= set(per)
    assert 0.9e9 < total < 1.3e9                                # 1.76 x the algorithmic 606 MB (profiles/round5/hbm_traffic_pmc.md)
    assert b.chain_traffic(4096, 256_000, 12345, False, True)[0] is None          # a configuration that was never profiled: no figure


def test_both_roofs_and_which_one_the_configuration_is_under():
    b = _bench()
    s = b.roof_sides(256_000, False, 4096 * 16384, 0.25e-3)             # configs[2]: 385 flop for 9.04 B per sample
    assert 42 < s["flop_per_byte"] < 43 and 19 < s["ridge_fp32_vector"] < 20 and 100 < s["ridge_mfma_bf16x3"] < 110
    assert s["bound"] == "hbm"                                          # above the vector ridge, below that of the matrix cores the FIRs run on
    assert abs(s["hbm"]["frac"] - 0.3032) < 1e-3 and abs(s["fp32_vector"]["frac"] - 0.657) < 1e-3
    assert abs(s["mfma_bf16x3"]["frac"] - 0.657 * 157.3 / (2500.0 / 3)) < 1e-3
    assert s["hbm"]["unit"] == "GB/s" and s["fp32_vector"]["unit"] == "TFLOP/s"
    assert b.roof_sides(1_024_000, True, 1.0, 1.0)["bound"] == "mfma"   # u8 at 1.024 MSa/s: 2.3 B per sample under the same flops


def test_the_traffic_table_is_tied_to_the_library_it_was_measured_on():
    b = _bench()
    stamp = b.kernel_source_stamp()
    assert len(stamp) == 16 and stamp == b.kernel_source_stamp()
    meta = b.traffic_meta()
    # the committed table carries the stamp of the build its rows come from (tools/digest_round.py --install); bench.py prints
    # traffic_stale = (that stamp != this tree's)
    assert "kernel_source_stamp" in meta and len(meta["kernel_source_stamp"]) == 16
