"""Phase split of the per-round NW kernel (k_nw_ad) on a fixed batch: one b_compare round against centre 0 of a synthetic
sample through dada2hip_sample_compare (event-timed NW launch), the batch thinned to a target size with the skip mask,
timed with DADA2HIP_AD_DEBUG phase-skipping bits (1 DP, 2 traceback, 4 factors, 8 product; results are void then).

Needs the profiling build of the library (`make -C dada2_amd/csrc prof`: the phase-skipping knob is compiled out of
libdada2hip.so).

    python tools/nw_phases.py [--config 3|5] [--uniques 300000] [--sizes 2000,4000,8700,18000,36000]"""
import argparse
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, help="bench.py configuration whose geometry is profiled (5: long reads, band 32)")
    ap.add_argument("--uniques", type=int, default=300000)
    ap.add_argument("--sizes", default="2000,4000,8700,18000,36000")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from dada2_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "dada2_amd", "libdada2hip_prof.so")
    from dada2_amd import api
    from dada2_amd.opts import DadaOpts
    args = types.SimpleNamespace(uniques=a.uniques, length=0, variants=0, deep=False)
    dereps, inputs, err, mine, c = bench.make_inputs(a.config, args, 0)
    d = dereps[0]
    opts = DadaOpts(BAND_SIZE=c["band"])
    s = api.Sample.from_derep(d, device=0)
    os.environ["DADA2HIP_NW_KERNEL"] = "coop"
    lam, ham, cls, st = s.compare(0, err, opts)
    nw_idx = np.flatnonzero(cls == 3)
    print(json.dumps({"config": a.config, "uniques": d.nraw, "maxlen": max(map(len, d.seqs)), "band": c["band"], "nw_candidates_of_centre0": int(nw_idx.size)}), flush=True)
    rng = np.random.default_rng(1)
    for size in [int(x) for x in a.sizes.split(",")]:
        if size > nw_idx.size:
            continue
        keep = rng.choice(nw_idx, size=size, replace=False)
        skip = np.ones(d.nraw, dtype=np.uint8)
        skip[keep] = 0
        for _once in (0,):
            row = {"batch": size}
            for bits in (0, 2, 3, 4, 8, 15):   # (bit 1 alone would walk garbage pointers and trip the range flag)
                os.environ["DADA2HIP_AD_DEBUG"] = str(bits)
                t = []
                for _ in range(a.reps):
                    _, _, _, st = s.compare(0, err, opts, skip=skip)
                    t.append(st["nw_kernel_ms"] * 1e3)
                assert st["nnw"] == size, (st["nnw"], size)
                row[f"us_skip{bits}"] = round(min(t), 1)
            os.environ["DADA2HIP_AD_DEBUG"] = "0"
            row["dp_us"] = round(row["us_skip2"] - row["us_skip3"], 1)
            row["traceback_us"] = round(row["us_skip0"] - row["us_skip2"], 1)
            row["factors_us"] = round(row["us_skip0"] - row["us_skip4"], 1)
            row["product_us"] = round(row["us_skip0"] - row["us_skip8"], 1)
            row["floor_us"] = row["us_skip15"]
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
