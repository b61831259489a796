"""oracle/derep.py - CPU restatement of the reference's dereplication (TEST INFRASTRUCTURE ONLY).

Checker of the product's host-side dereplicator ``dada2hip_derep_fastq`` (dada2_amd/csrc/derep.cpp).  Only ``tests/``
and the golden generators under ``tests/golden/`` import it; nothing under ``dada2_amd/`` does.

What it follows, line by line (/root/reference/R/sequenceIO.R):
  * ``derepFastq`` :45-124 - ``FastqStreamer(fl, n)`` + ``yield`` (:56-57): the file is consumed in chunks of ``n``
    records; ``qtables2`` per chunk (:59, :72); uniques already seen are summed in place (:80-84), new ones are appended
    BEHIND the earlier ones (:85-88); the read map of a later chunk is re-based by ``match`` (:90-92); means =
    ``derepQuals / derepCounts`` (:95); final order = ``order(derepCounts, decreasing=TRUE)`` (:98), which is stable,
    so ties keep the order above; the map follows by ``match(derepMap, ord)`` (:101).
  * ``qtables2`` :150-183 - zero-length reads are dropped from the chunk (:154-158) and their map entries are NA
    (:173-177); ``srsort`` (:161) puts the chunk's reads in C-locale lexical order, so the chunk's uniques come out
    in that order (:163-165); ``cum_quals`` = per-unique, per-position SUM of the integer quality scores (:166-171),
    NA past the end of a short unique (the matrix is ``width``-wide, :169).
  * quality decoding: ``qualityType="Auto"`` (:45, ShortRead): Phred+33 unless no character below ';' (59) occurs,
    then Phred+64.  ShortRead is an R package absent from /root/reference: parity with genuine ShortRead output is
    UNPINNED here; what IS pinned is (a) the uniques / abundances / order of the reference's own fixtures against
    coreutils ``sort | uniq -c`` (tests/test_derep.py), (b) the known unique counts of SURVEY.md §4/§6.
"""
from __future__ import annotations

import gzip

import numpy as np

from dada2_amd.io import Derep


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
    """derepFastq of one file (small files only)."""
    s, q = read_fastq(path)
    return derep_from_reads(s, q, n)
