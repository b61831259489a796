"""Tuning sweep: one synthetic sample (bench.py's configuration N) made once and kept resident, then timed under a list
of environment settings (the library reads its DADA2HIP_* knobs at every run).  One JSON line per setting with the wall
time of a resident pass and the event-timed device milliseconds per phase.

    python tools/sweep_env.py --config 3 --reps 3 "A=1" "A=2 B=3" ...
The first (implicit) setting is the default environment; every result is checked against it (same partitions / map)."""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--uniques", type=int, default=0)
    ap.add_argument("--deep", action="store_true", help="bench.py's --deep variant of the configuration (28 reads per unique at config 2)")
    ap.add_argument("settings", nargs="*")
    ap.add_argument("--list", default="", help="settings separated by ';'")
    a = ap.parse_args()
    if os.environ.get("DADA2HIP_LIB"):   # (a build variant of the library: code-generation experiments)
        from dada2_amd import _lib
        _lib.LIB_PATH = os.environ["DADA2HIP_LIB"]
    from dada2_amd import api
    from dada2_amd.opts import DadaOpts
    args = types.SimpleNamespace(uniques=a.uniques, length=0, variants=0, deep=a.deep)
    dereps, inputs, err, mine, c = bench.make_inputs(a.config, args, 0)
    d = dereps[0]
    opts = DadaOpts(BAND_SIZE=c["band"])
    s = api.Sample.from_derep(d, device=0)
    base = None
    for setting in [""] + list(a.settings) + [x.strip() for x in a.list.split(";") if x.strip()]:
        kv = dict(x.split("=", 1) for x in setting.split())
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        os.environ["DADA2HIP_PROFILE"] = "0"
        walls = []
        per_pass = []          # what the library itself timed in each pass (a slow pass: where?)
        res = None
        for _ in range(a.reps):
            t0 = time.perf_counter()
            res = s.run(err, opts)
            walls.append((time.perf_counter() - t0) * 1e3)
            rs = res.stats
            per_pass.append({"wall": round(walls[-1], 1), "total": round(rs["ms_total"], 1), "bookkeep": round(rs["ms_bookkeep"], 1), "wait": round(rs["ms_wait_device"], 1),
                             "replay": round(rs["ms_replay"], 1), "enqueue": round(rs["ms_enqueue"], 1), "final": round(rs["ms_final"], 1),
                             "setup": round(rs["ms_setup"], 1), "round0": round(rs["ms_round0"], 1), "upload": round(rs["ms_upload"], 1),
                             "launches": int(rs["tail_launches"]), "pauses": int(rs["tail_pauses"]), "pf_waits": int(rs["pf_waits"]), "pf_exits": int(rs["pf_exits"]),
                             "compares": int(rs["batch_compares"]), "pf": int(rs["pf_compares"])})
        os.environ["DADA2HIP_PROFILE"] = "1"
        prof = s.run(err, opts)
        st = prof.stats
        same = True
        if base is None:
            base = res
        else:
            same = bool(np.array_equal(base.map, res.map) and np.array_equal(base.clustering["abundance"], res.clustering["abundance"]))
        print(json.dumps({"setting": setting or "(default)", "ms_min": round(min(walls), 2), "ms_all": [round(w, 2) for w in walls],
                          "same_as_default": same, "nclust": int(res.nclust),
                          "dev_ms": {k[7:]: round(v, 2) for k, v in st.items() if k.startswith("dev_ms_")},
                          "tail_block0_ms": {k[8:]: round(v, 2) for k, v in st.items() if k.startswith("tail_ms_")},
                          "overlap": {k: int(st[k]) for k in ("pf_compares", "pf_centres", "pf_hits", "pf_waits", "pf_exits", "batch_compares", "tail_launches")},
                          "aligned_in_vain_frac": (round(1.0 - st["nnw_rounds"] / st["nnw_run"], 4) if st["nnw_run"] else None),
                          "nnw": int(st["nnw"]), "nnw_run": int(st["nnw_run"]), "nnw_fast": int(st["nnw_fast"]), "nnw_retry": int(st["nnw_retry"]),
                          "screen_stage2": int(st["screen_stage2"]), "batch_compares": int(st["batch_compares"]),
                          "wait_device": round(st.get("ms_wait_device", 0), 1), "replay": round(st.get("ms_replay", 0), 1),
                          "per_pass": per_pass}), flush=True)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == "__main__":
    main()
