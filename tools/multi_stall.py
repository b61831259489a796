"""tools/multi_stall.py - development aid: dada2hip_run_multi with four samples of 60 k uniques, two in flight on one GPU, in fresh
processes; prints every run's wall time and, per sample, where the library says the time went (which run stalled, and where)."""
import json
import os
import pickle
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

CHILD = r"""
import sys, time, json, pickle
import numpy as np
root = %r
sys.path[:0] = [root, root + '/tests']
from helpers import tperr1
from dada2_amd import api
from dada2_amd.io import Derep
from dada2_amd.opts import DadaOpts
seqs, ab, q = pickle.load(open(%r, 'rb'))
dereps = []
for i in range(4):
    a = ab.copy(); a[: 40 * i] += 1
    dereps.append(Derep(seqs, a, q, np.zeros(0, np.int32)))
his = [api.HostInput.from_derep(d) for d in dereps]
api.dada_uniques_multi(his[:2], tperr1(), DadaOpts(), devices=(0, 0))
t0 = time.perf_counter()
res = api.dada_uniques_multi(his, tperr1(), DadaOpts(), devices=(0, 0))
ms = (time.perf_counter() - t0) * 1e3
keys = ('ms_total', 'ms_upload', 'ms_setup', 'ms_round0', 'ms_bookkeep', 'ms_wait_device', 'ms_replay', 'ms_enqueue', 'ms_final', 'tail_launches', 'tail_fallbacks', 'overlap_on', 'lite_misses', 'batch_compares')
print(json.dumps({'ms': ms, 'nclust': [int(r.nclust) for r in res], 'stats': [{k: round(float(r.stats[k]), 1) for k in keys} for r in res]}))
"""


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    from helpers import tperr1
    from dada2_amd.synth import make_sample
    d0 = make_sample(tperr1(), 60000, L=250, G=48, seed=5100, chunk=60000)
    path = "/tmp/multi_stall_sample.pkl"
    with open(path, "wb") as fh:
        pickle.dump((d0.seqs, d0.abundances, d0.quals), fh, protocol=4)
    walls = []
    for k in range(n):
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, path)], capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print("run", k, "failed", out.stderr[-2000:])
            continue
        r = json.loads(out.stdout.strip().splitlines()[-1])
        walls.append(r["ms"])
        med = sorted(walls)[len(walls) // 2]
        rec = {"run": k, "ms": round(r["ms"], 1), "ms_total_per_sample": [s["ms_total"] for s in r["stats"]]}
        if r["ms"] > 1.4 * med:
            rec["stats"] = r["stats"]                      # (a slow run: everything the library timed)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
