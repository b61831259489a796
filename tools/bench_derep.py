#!/usr/bin/env python3
"""Throughput of the dereplication front-end (dada2hip_derep_fastq, host-side C++; SURVEY.md §8f rank 3: R's derepFastq takes
minutes at 10^6 uniques).  Writes a synthetic FASTQ (plain and gzip) with bench.py's read recipe, then times the call.
No GPU involved.  Usage: bench_derep.py [n_reads=1200000] [L=250]"""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 250
rng = np.random.default_rng(11)
G = 2048
anc = rng.integers(0, 4, size=L, dtype=np.uint8)
tv = np.tile(anc, (G, 1))
for g in range(G):
    p = rng.choice(L, size=int(rng.integers(1, 40)), replace=False)
    tv[g, p] = (tv[g, p] + rng.integers(1, 4, size=p.size, dtype=np.uint8)) & 3
w = np.arange(1, G + 1, dtype=np.float64) ** -1.1
w /= w.sum()
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
tmp = tempfile.mkdtemp(prefix="dada2hip_derep_")
plain = os.path.join(tmp, "reads.fastq")
t0 = time.perf_counter()
with open(plain, "wb") as fh:
    for lo in range(0, n_reads, 100_000):
        n = min(100_000, n_reads - lo)
        codes = tv[rng.choice(G, size=n, p=w)].copy()
        q = np.clip(np.rint(np.linspace(38, 22, L)[None, :] + rng.normal(0, 4, size=(n, L))), 2, 40).astype(np.uint8)
        err = rng.random(size=(n, L)) < 10.0 ** (-q / 10.0)
        codes[err] = (codes[err] + rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)) & 3
        seq = lut[codes]
        qs = (q + 33).astype(np.uint8)
        for i in range(n):
            fh.write(b"@r%d\n" % (lo + i) + seq[i].tobytes() + b"\n+\n" + qs[i].tobytes() + b"\n")
t_write = time.perf_counter() - t0
gz = plain + ".gz"
with open(plain, "rb") as fi, gzip.open(gz, "wb", compresslevel=4) as fo:
    while True:
        b = fi.read(1 << 24)
        if not b:
            break
        fo.write(b)
from dada2_amd import api
out = {"n_reads": n_reads, "L": L, "cores": os.cpu_count(), "fastq_bytes": os.path.getsize(plain), "gz_bytes": os.path.getsize(gz)}
for name, path in (("plain", plain), ("gzip", gz)):
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        nd = api.NativeDerep(path, n=10**6)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        nu, nr = nd.nuniques, nd.nreads
        nd.close() if hasattr(nd, "close") else None
    out[name] = {"seconds": best, "reads_per_s": nr / best, "uniques": nu, "reads": nr, "MB_per_s": os.path.getsize(path) / best / 1e6}
os.remove(plain); os.remove(gz); os.rmdir(tmp)
print(json.dumps(out))
