"""bench.py's one-line JSON contract, checked on the committed records of the last GPU runs (profiles/r07*_bench_*.json):
the driver and the judge parse these keys, so a refactor of bench.py must keep them."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = ["r11d_bench_cfg3_default_line.json", "r11g_bench_cfg3_default_line_second_box.json", "r10x_bench_cfg3_default_line.json", "r10m_bench_cfg3_default_line.json", "r10f_bench_cfg3_default_line.json", "r08e_bench_cfg3_default_line_as_the_driver_runs_it.json", "r07w_bench_cfg3_default_line.json", "r08c_bench_cfg3_default_line_final_code_slow_host_box.json", "r07n_bench_cfg2.json", "r07i_bench_cfg4_inflight2_without.json", "r07n_bench_cfg5.json"]   # [0] = the default run


@pytest.mark.parametrize("name", RECORDS)
def test_record_has_the_contract_fields(name):
    d = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "uniques/s" and d["higher_is_better"] is True and d["data"] == "synthetic"
    # one sample per rank: per-GPU work fixed as N grows; configs[3] is a fixed pool of 8 samples dealt over the ranks
    assert d["scaling"] == ("strong" if d["config"]["baseline_config"] == 4 else "weak")
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if name == RECORDS[0]:
        sat = d["roofline_nw_saturated"]                  # one saturating launch of the per-round NW kernel
        assert sat["alignments"] >= 30000 and abs(sat["frac"] - sat["achieved"] / sat["peak"]) < 1e-9
    c = d["cpu_baseline"]
    if name == RECORDS[0]:
        assert c is not None                              # the default run always times the reference beside the GPU
    if c is not None:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("reference", "port")
        assert c.get("parity_vs_gpu", True) is True      # the reference run on the same input gave the same partitions / map
    # throughput is the whole job (all samples of all ranks) over the timed region
    cfg = d["config"]
    assert abs(d["value"] - cfg["uniques_per_sample"] * cfg["samples_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_metric_matches_baseline_json():
    b = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    d = json.load(open(os.path.join(ROOT, "profiles", RECORDS[0])))
    assert "unique" in d["metric"] and "unique" in json.dumps(b).lower()
    # the default run is the configuration the metric is quoted on: 1M uniques x 250 nt (configs[2]), timed around the
    # whole boundary call (host buffers in, results out)
    assert d["config"]["baseline_config"] == 3 and d["config"]["uniques_per_sample"] == 1_000_000
    assert "boundary" in json.dumps(d["config"]).lower() or "boundary" in d.get("timed_region", "").lower()


def test_default_line_names_the_dominant_kernel_class_and_carries_the_sub_records():
    """`roofline` is the class with the largest device time of the event-timed pass - the persistent round tail - not the
    larger of aligner and screen; the whole-sample reference run is the top level of cpu_baseline; configs[3] / [4] ride in the line."""
    d = json.load(open(os.path.join(ROOT, "profiles", RECORDS[0])))
    dev = d["phases_ms_last_step"]["device_ms_profiled_pass"]
    assert d["roofline"]["kernel"].startswith("k3_tail") and d["roofline"]["kernel_ms"] == pytest.approx(dev["tail"])
    assert d["roofline"]["kernel_ms"] >= max(r["kernel_ms"] for r in d["roofline_others"])
    assert d["roofline"]["latency_model"]["rounds"] == d["config"]["partitions"] - 1
    c = d["cpu_baseline"]
    assert "WHOLE" in c["sample"] and c["parity_vs_gpu"] is True and c["prefix"]["value"] > 0
    assert d["config5_long_reads"]["steps"] == 2 and d["config5_long_reads"]["partitions"] > 100
    assert d["config4_eight_samples_one_gpu"]["samples"] == 8
    assert d["per_rank"][0]["samples"] == 1 and d["samples_in_flight_per_gpu"] == 1
    # round 6: the plain configs[1] sample rides in the line, the tail's roofline says that HBM is not its
    # bound (PMC traffic / algorithmic bytes), and the aligned-in-vain share is reported (and is a share: round 0's pairs no
    # longer count as committed by the rounds)
    assert d["config2_plain_100k"]["steps"] == 3 and d["config2_plain_100k"]["partitions"] > 100
    assert 0 < d["roofline"]["hbm_is_not_the_bound"]["traffic_over_algorithmic"] < 1
    assert 0 <= d["phases_ms_last_step"]["alignments"]["aligned_in_vain_frac"] <= 0.12
    assert 0 <= d["secondary_workload"]["aligned_in_vain_frac"] <= 0.12
