"""Host-side containers and helpers around the hot path's input (no GPU work here).

* ``Derep`` mirrors the R ``derep-class`` object that ``dada()`` consumes; the product's dereplicator is
  ``dada2hip_derep_fastq`` (csrc/derep.cpp, reached through ``dada2_amd.api.derep_fastq``).  The Python restatement
  of ``derepFastq`` that CHECKS it lives under ``oracle/derep.py`` (test infrastructure, not in this package).
* the bundled error matrices ``data/{tperr1,errBalancedF,errBalancedR}.rda`` (gzip'd RDX2/XDR; 16 x 41 doubles, rows
  A2A..T2T, columns Q0..Q40) and the two error-matrix helpers of R/dada.R and R/errorModels.R.
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

    def qmax(self) -> int:
        """``ceiling(max(derep$quals, na.rm=TRUE))`` (R/dada.R:307), taken once per object: the quality matrix is 2 GB at 10^6
        uniques x 250 nt, and the pass loop of ``dada()`` asked for it in every pass."""
        q = self.__dict__.get("_qmax")
        if q is None:
            q = int(np.ceil(np.nanmax(self.quals)))
            self.__dict__["_qmax"] = q
        return q


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
