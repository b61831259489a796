"""Producers of the hot path's input, restated host-side (no GPU work here).

The reference builds ``dada_uniques``' arguments in R:

* ``derepFastq`` / ``qtables2``  (/root/reference/R/sequenceIO.R:45-124, :150-183):
  uniques in C-locale lexical order (ShortRead ``srsort``), per-unique mean quality
  (``derepQuals/derepCounts`` :95), then a *stable* sort by decreasing abundance (:98).
* the bundled error matrices ``data/{tperr1,errBalancedF,errBalancedR}.rda``
  (gzip'd RDX2/XDR; 16 x 41 doubles, rows A2A..T2T, columns Q0..Q40).

ShortRead/Biostrings are R packages and are not available here, so these ~60 lines
restate just what the parity tests and ``bench.py`` need to reproduce the same inputs.
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass

import numpy as np


@dataclass
class Derep:
    """Mirror of the R ``derep-class`` list (R/allClasses.R): uniques (name=sequence,
    value=abundance), quals (one ROW per unique, NaN past a short read's end), map."""

    seqs: list          # list[str], abundance-sorted
    abundances: np.ndarray  # int32 [N]
    quals: np.ndarray   # float64 [N, maxlen]  (R layout; dada() transposes, R/dada.R:337)
    map: np.ndarray     # int32 [nreads] 0-based index into seqs

    @property
    def nraw(self) -> int:
        return len(self.seqs)


def read_fastq(path: str):
    """Minimal 4-line FASTQ reader -> (list[str] seqs, list[bytes] quals). Phred+33."""
    op = gzip.open if str(path).endswith(".gz") else open
    seqs, quals = [], []
    with op(path, "rb") as fh:
        while True:
            h = fh.readline()
            if not h:
                break
            s = fh.readline().rstrip(b"\r\n")
            fh.readline()
            q = fh.readline().rstrip(b"\r\n")
            seqs.append(s.decode("ascii"))
            quals.append(q)
    return seqs, quals


def derep_from_reads(seqs, quals_phred33, n: int = 10**6, offset: int = 33) -> Derep:
    """qtables2 per chunk of ``n`` reads + derepFastq's merge and tail (sequenceIO.R:150-183, :57-101): inside a chunk
    the new uniques are in C-locale lexical order, later chunks append theirs (:85-88)."""
    order_seen, count, qsum = [], {}, {}
    for c0 in range(0, len(seqs), n):
        chunk = range(c0, min(c0 + n, len(seqs)))
        fresh = sorted({seqs[i] for i in chunk if len(seqs[i]) > 0 and seqs[i] not in count})  # srsort: C-locale order
        for s in fresh:
            order_seen.append(s)
            count[s] = 0
            qsum[s] = np.zeros(len(s))
        for i in chunk:
            s = seqs[i]
            if len(s) == 0:  # zero-length reads are ignored (:154-158)
                continue
            count[s] += 1
            qsum[s] += np.frombuffer(quals_phred33[i], dtype=np.uint8).astype(np.float64) - float(offset)
    if not order_seen:
        raise ValueError("Only zero-length sequences detected during dereplication.")
    uniq = order_seen
    maxlen = max(len(s) for s in uniq)
    counts = np.array([count[s] for s in uniq], dtype=np.int64)
    cum = np.full((len(uniq), maxlen), np.nan)
    for u, s in enumerate(uniq):
        cum[u, : len(s)] = qsum[s]
    mean = cum / counts[:, None]                      # derepQuals/derepCounts (:95)
    order = np.argsort(-counts, kind="stable")        # order(derepCounts, decreasing=TRUE) (:98), stable
    rank_of = np.empty(len(uniq), dtype=np.int64)
    rank_of[order] = np.arange(len(uniq))
    uidx = {s: u for u, s in enumerate(uniq)}
    rmap = np.full(len(seqs), -1, dtype=np.int32)
    for i, s in enumerate(seqs):
        if len(s) > 0:
            rmap[i] = rank_of[uidx[s]]
    return Derep([uniq[u] for u in order], counts[order].astype(np.int32), mean[order], rmap)


def derep_fastq(path: str, n: int = 10**6) -> Derep:
    """Python restatement (the checker of dada2hip_derep_fastq; small files only)."""
    s, q = read_fastq(path)
    return derep_from_reads(s, q, n)


TRANS_NAMES = [a + "2" + b for a in "ACGT" for b in "ACGT"]  # A2A, A2C, ..., T2T (R/dada.R:362)


def load_err_rda(path: str) -> np.ndarray:
    """Decode a 16 x Q error matrix from one of the reference's ``data/*.rda`` files.

    The serialized object is a REALSXP of length 16*Q written big-endian right after its
    int32 length; locate that length word and read 16*Q doubles (column-major)."""
    raw = gzip.open(path, "rb").read()
    assert raw[:5] == b"RDX2\n", "not an RDX2 .rda"
    for ncol in (41, 94, 42, 43):
        n = 16 * ncol
        tag = b"\x00\x00\x02\x0e" + int(n).to_bytes(4, "big")  # REALSXP (type 14, has-attr) + length
        i = raw.find(tag)
        if i >= 0:
            a = np.frombuffer(raw[i + 8 : i + 8 + 8 * n], dtype=">f8").astype(np.float64)
            return np.ascontiguousarray(a.reshape(ncol, 16).T)
    raise ValueError("no 16xQ REALSXP found in " + str(path))


def extend_err(err: np.ndarray, qmax: int) -> np.ndarray:
    """R/dada.R:303-313 — repeat the last column until the matrix covers 0..qmax."""
    err = np.asarray(err, dtype=np.float64)
    while err.shape[1] < qmax + 1:
        err = np.concatenate([err, err[:, -1:]], axis=1)
    return err


def inflate_err(err: np.ndarray, inflation: float, inflate_self: bool = False) -> np.ndarray:
    """R/errorModels.R:446-456 (inflateErr)."""
    err = np.array(err, dtype=np.float64, copy=True)
    rows = [r for r in range(16) if (r % 5 != 0) or inflate_self]
    err[rows, :] = (err[rows, :] * inflation) / (1 + (inflation - 1) * err[rows, :])
    return err
