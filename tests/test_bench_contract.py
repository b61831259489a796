"""bench.py's one-line JSON contract, checked on the committed records of the last GPU runs (profiles/r01_bench_*.json):
the driver and the judge parse these keys, so a refactor of bench.py must keep them."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = ["r01_bench_default.json", "r01_bench_1M_with_cpu_reference.json", "r01_bench_long_reads_20k_x_1500.json"]


@pytest.mark.parametrize("name", RECORDS)
def test_record_has_the_contract_fields(name):
    d = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "uniques/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # throughput is the whole job over the timed region
    assert abs(d["value"] - d["config"]["uniques_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert c.get("parity_vs_gpu", True) is True          # the reference run on the same input gave the same partitions / map


def test_bench_metric_matches_baseline_json():
    b = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    d = json.load(open(os.path.join(ROOT, "profiles", RECORDS[0])))
    assert "unique" in d["metric"] and "unique" in json.dumps(b).lower()
    assert d["config"]["uniques_per_gpu"] == 100_000   # configs[1]: the single-GPU configuration the metric is quoted on
