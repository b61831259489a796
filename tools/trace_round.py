#!/usr/bin/env python3
"""tools/trace_round.py - phase timeline of the round-tail kernels of a few rounds of the headline run.

Every block of k2_shuffle (levels 0..3) / k2_pupdate / k2_birth stamps the shader clock (s_memtime) at its
phase boundaries when DADA2HIP_V2_TRACE=<block sequence number>:<file> is set (Eng2::trace, rounds2.inc.hip).  This
script runs resident passes of bench.py's sample with the trace on for the requested rounds and prints, per kernel: the
span from the first block's start to the last block's end, the skew of the block starts, and the time the blocks spend in
each phase (mean / max).  usage: python tools/trace_round.py [--config 3] [--seqs 120,300,500] > profiles/<tag>_round_trace.json"""
import argparse
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNELS = ["(k2_lists: folded into the store pass in round 3, no stamps)", "k2_shuffle L0 (commit + store)", "k2_shuffle L1", "k2_shuffle L2", "k2_shuffle L3", "k2_pupdate", "k2_birth"]
PHASES = {0: ["entry->lists built", "lists written"], 1: ["prologue (tables)", "main loop", "buffers out", "stats", "deltas out"],
          5: ["prologue (tables)", "main loop", "reduce", "sig list out"],
          6: ["fold deltas", "arg-min + ties", "decision", "flags", "birth + plan", "publish"]}
PHASES[2] = PHASES[3] = PHASES[4] = PHASES[1]
TB = 4096


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--seqs", default="60,300,600")
    ap.add_argument("--clock-mhz", type=float, default=2400.0, help="s_memtime tick rate in MHz (MI355X_MICROARCH.md: one tick per shader cycle; nominal clock)")
    a = ap.parse_args()
    from dada2_amd import api
    from dada2_amd.opts import DadaOpts
    args = types.SimpleNamespace(uniques=0, length=0, variants=0, deep=False)
    dereps, inputs, err, mine, c = bench.make_inputs(a.config, args, 0)
    opts = DadaOpts(BAND_SIZE=c["band"])
    s = api.Sample.from_derep(dereps[0], device=0)
    s.run(err, opts)
    out = []
    for seq in [int(x) for x in a.seqs.split(",")]:
        path = f"/tmp/d2trace_{seq}.bin"
        os.environ["DADA2HIP_V2_TRACE"] = f"{seq}:{path}"
        os.environ["DADA2HIP_V2_GRAPH"] = "1"
        r = s.run(err, opts)
        t = np.fromfile(path, dtype=np.uint64).reshape(8, TB, 8).astype(np.float64)
        rec = {"round_block_seq": seq, "partitions_total": int(r.nclust), "kernels": {}}
        t00 = None
        for k, name in enumerate(KERNELS):
            blk = t[k]
            used = blk[:, 0] > 0
            if not used.any():
                continue
            b = blk[used]
            nph = len(PHASES[k])
            start, end = b[:, 0], b[:, nph]
            ok = end > 0
            if t00 is None:
                t00 = start.min()
            tick_us = 1.0 / a.clock_mhz
            e = {"blocks": int(used.sum()), "first_start_us_since_round_start": round((start.min() - t00) * tick_us, 2),
                 "span_us": round((end[ok].max() - start.min()) * tick_us, 2) if ok.any() else None,
                 "start_skew_us": round((start.max() - start.min()) * tick_us, 2), "phases_us_mean_max": {}}
            for p, pn in enumerate(PHASES[k]):
                d = (b[:, p + 1] - b[:, p])[(b[:, p + 1] > 0) & (b[:, p] > 0)] * tick_us
                if d.size:
                    e["phases_us_mean_max"][pn] = [round(float(d.mean()), 2), round(float(d.max()), 2)]
            rec["kernels"][name] = e
        out.append(rec)
        print(json.dumps(rec), flush=True)
    del os.environ["DADA2HIP_V2_TRACE"]
    s.close()


if __name__ == "__main__":
    main()
