#!/usr/bin/env python3
"""Timing of the bimera step on a synthetic sequence table: dada2hip_table_bimera2 on the GPU vs the
reference's C_table_bimera2 (oracle/_ref, all host cores), same table, results compared.  Usage: bench_bimera.py [nseq] [nsam] [L]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_table(nseq=3000, nsam=8, L=250, seed=7):
    """A sequence table of the shape removeBimeraDenovo sees: nseq / 4 true sequences at 3-15 % divergence from one ancestor,
    abundant in most samples, the rest two-parent mosaics of them at low abundance.  Returns (mat [nsam, nseq], seqs)."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, size=L)
    true = []
    for g in range(nseq // 4):
        s = anc.copy()
        p = rng.choice(L, size=int(L * rng.uniform(0.03, 0.15)), replace=False)
        s[p] = (s[p] + rng.integers(1, 4, size=p.size)) & 3
        true.append("".join("ACGT"[x] for x in s))
    seqs = list(dict.fromkeys(true))
    nt = len(seqs)
    seen = set(seqs)
    while len(seqs) < nseq:
        a, b = rng.choice(nt, 2, replace=False)
        cut = int(rng.integers(10, L - 10))
        ch = seqs[a][:cut] + seqs[b][cut:]
        if ch not in seen:
            seen.add(ch); seqs.append(ch)
    mat = np.zeros((nsam, len(seqs)), dtype=np.int32)
    mat[:, :nt] = rng.integers(0, 2000, size=(nsam, nt)) * (rng.random((nsam, nt)) < 0.7)
    mat[:, nt:] = rng.integers(0, 20, size=(nsam, len(seqs) - nt)) * (rng.random((nsam, len(seqs) - nt)) < 0.5)
    return mat, seqs


def record(api, nseq=3000, nsam=8, L=250, device=0, reference=True):
    """One timed dada2hip_table_bimera2 call on make_table(); with `reference` the reference's C_table_bimera2 (oracle/_ref, all
    host cores) on the same table beside it, results compared."""
    mat, seqs = make_table(nseq, nsam, L)
    api.table_bimera2(mat[:, :64], seqs[:64], device=device)          # warm-up (context, allocation cache)
    t0 = time.perf_counter()
    got = api.table_bimera2(mat, seqs, device=device)
    t_gpu = time.perf_counter() - t0
    out = {"what": "C_table_bimera2 (src/chimera.cpp:192; SURVEY.md 8f rank 2) on a synthetic sequence table", "nseq": len(seqs),
           "nsam": nsam, "L": L, "gpu_s": t_gpu, "flagged": int((got[0] > 0).sum())}
    if reference:
        from oracle import ref
        if ref.available():
            ref.set_threads(os.cpu_count() or 1)
            t0 = time.perf_counter()
            want = ref.table_bimera2(mat, seqs)
            out["reference_all_cores_s"] = time.perf_counter() - t0
            out["cores"] = os.cpu_count()
            out["equal"] = bool(np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]))
            out["speedup"] = out["reference_all_cores_s"] / t_gpu
    return out


if __name__ == "__main__":
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    nsam = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 250
    from dada2_amd import api
    print(json.dumps(record(api, nseq, nsam, L, reference=not os.environ.get("BIMERA_NO_REF"))))
