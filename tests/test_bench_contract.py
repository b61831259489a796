"""bench.py's one-line JSON contract, checked on the committed records of the last GPU runs (profiles/r05z_bench_*.json):
the driver and the judge parse these keys, so a refactor of bench.py must keep them."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = ["r05z_bench_cfg3.json", "r05z_bench_cfg2.json", "r05z_bench_cfg4_inflight2.json", "r05z_bench_cfg5.json"]   # [0] = the default run


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
