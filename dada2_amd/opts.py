"""Option store and result container for the ``dada_uniques`` boundary.

``DadaOpts`` mirrors the reference's ``dada_opts`` environment and its defaults
(/root/reference/R/dada.R:1-27) and the normalisation ``dada()`` applies before the
``.Call`` (R/dada.R:222-237).  ``COpts`` is the ctypes image of ``dada2hip_opts``
(include/dada2hip.h) — the 23 scalars ``dada_uniques`` takes positionally
(/root/reference/src/Rmain.cpp:33-47).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field, asdict
from typing import Optional

import numpy as np

NA_INTEGER = -(2 ** 31)  # R's NA_integer_


class COpts(C.Structure):
    _fields_ = [
        ("kdist_cutoff", C.c_double), ("omegaA", C.c_double), ("omegaP", C.c_double),
        ("omegaC", C.c_double), ("min_fold", C.c_double),
        ("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32), ("homo_gap", C.c_int32),
        ("band_size", C.c_int32), ("max_clust", C.c_int32), ("min_hamming", C.c_int32),
        ("min_abund", C.c_int32),
        ("use_kmers", C.c_int32), ("detect_singletons", C.c_int32), ("use_quals", C.c_int32),
        ("final_consensus", C.c_int32), ("vectorized_alignment", C.c_int32),
        ("multithread", C.c_int32), ("verbose", C.c_int32), ("SSE", C.c_int32),
        ("gapless", C.c_int32), ("greedy", C.c_int32),
    ]


assert C.sizeof(COpts) == 112


@dataclass
class DadaOpts:
    OMEGA_A: float = 1e-40
    OMEGA_P: float = 1e-4
    OMEGA_C: float = 1e-40
    DETECT_SINGLETONS: bool = False
    USE_KMERS: bool = True
    KDIST_CUTOFF: float = 0.42
    MAX_CONSIST: int = 10
    MATCH: int = 5
    MISMATCH: int = -4
    GAP_PENALTY: int = -8
    BAND_SIZE: int = 16
    VECTORIZED_ALIGNMENT: bool = True
    MAX_CLUST: int = 0
    MIN_FOLD: float = 1
    MIN_HAMMING: int = 1
    MIN_ABUNDANCE: int = 1
    USE_QUALS: bool = True
    HOMOPOLYMER_GAP_PENALTY: Optional[int] = None
    SSE: int = 2
    GAPLESS: bool = True
    GREEDY: bool = True
    PSEUDO_PREVALENCE: int = 2
    PSEUDO_ABUNDANCE: float = float("inf")

    def normalised(self) -> "DadaOpts":
        """R/dada.R:188-191,222-237: validate omegas, sign-fix gap penalties, homo_gap
        defaults to gap, vectorized aligner off for homopolymer gapping or BAND_SIZE 0."""
        o = DadaOpts(**asdict(self))
        if o.OMEGA_A < 0 or o.OMEGA_A >= 1:
            raise ValueError("OMEGA_A must be between zero and one.")
        if o.OMEGA_P < 0 or o.OMEGA_P >= 1:
            raise ValueError("OMEGA_P must be between zero and one.")
        if o.GAP_PENALTY > 0:
            o.GAP_PENALTY = -o.GAP_PENALTY
        if o.HOMOPOLYMER_GAP_PENALTY is None:
            o.HOMOPOLYMER_GAP_PENALTY = o.GAP_PENALTY
        if o.HOMOPOLYMER_GAP_PENALTY > 0:
            o.HOMOPOLYMER_GAP_PENALTY = -o.HOMOPOLYMER_GAP_PENALTY
        if o.HOMOPOLYMER_GAP_PENALTY != o.GAP_PENALTY:
            o.VECTORIZED_ALIGNMENT = False
        if o.VECTORIZED_ALIGNMENT and o.BAND_SIZE == 0:
            o.VECTORIZED_ALIGNMENT = False
        return o

    def to_c(self, *, max_clust=None, multithread=False, verbose=False) -> COpts:
        """The positional scalars of the .Call at R/dada.R:335-352 (use_quals hard-wired
        TRUE :344, final_consensus FALSE :345)."""
        o = self.normalised()
        return COpts(
            kdist_cutoff=float(o.KDIST_CUTOFF), omegaA=float(o.OMEGA_A), omegaP=float(o.OMEGA_P),
            omegaC=float(o.OMEGA_C), min_fold=float(o.MIN_FOLD),
            match=int(o.MATCH), mismatch=int(o.MISMATCH), gap=int(o.GAP_PENALTY),
            homo_gap=int(o.HOMOPOLYMER_GAP_PENALTY), band_size=int(o.BAND_SIZE),
            max_clust=int(o.MAX_CLUST if max_clust is None else max_clust),
            min_hamming=int(o.MIN_HAMMING), min_abund=int(o.MIN_ABUNDANCE),
            use_kmers=int(o.USE_KMERS), detect_singletons=int(o.DETECT_SINGLETONS), use_quals=1,
            final_consensus=0, vectorized_alignment=int(o.VECTORIZED_ALIGNMENT),
            multithread=int(bool(multithread)), verbose=int(bool(verbose)), SSE=int(o.SSE),
            gapless=int(o.GAPLESS), greedy=int(o.GREEDY),
        )


CLUSTERING_COLS = ["sequence", "abundance", "n0", "n1", "nunq", "pval", "birth_from", "birth_pval",
                   "birth_fold", "birth_ham", "birth_qave"]
BIRTH_SUBS_COLS = ["pos", "ref", "sub", "qual", "clust"]


@dataclass
class DadaResult:
    """The six objects ``dada_uniques`` returns (/root/reference/src/Rmain.cpp:294):
    clustering / birth_subs are column dicts (DataFrames in R), subqual is 16 x Q int32,
    clusterquals is maxlen x C float64, map is int32 [N] (1-based, NA_INTEGER where the
    unique is not corrected), pval is float64 [N]."""

    clustering: dict
    birth_subs: dict
    subqual: np.ndarray
    clusterquals: np.ndarray
    map: np.ndarray
    pval: np.ndarray
    stats: dict = field(default_factory=dict)

    @property
    def nclust(self) -> int:
        return len(self.clustering["sequence"])
