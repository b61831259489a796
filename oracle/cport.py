"""ctypes front-end to oracle/libdada2oracle.so — the plain-C restatement
(oracle/dada_oracle.c + oracle/rmath_ppois.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from dada2_amd.opts import COpts, DadaOpts, DadaResult
from oracle.ref import pack_inputs

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libdada2oracle.so")
_lib = None


class _CResult(C.Structure):
    _fields_ = [
        ("nclust", C.c_int), ("nraw", C.c_int), ("maxlen", C.c_int), ("ncol", C.c_int), ("nbirth_subs", C.c_int),
        ("nalign", C.c_uint), ("nshroud", C.c_uint),
        ("cl_sequence", C.POINTER(C.c_char_p)),
        ("cl_abundance", C.POINTER(C.c_int)), ("cl_n0", C.POINTER(C.c_int)), ("cl_n1", C.POINTER(C.c_int)),
        ("cl_nunq", C.POINTER(C.c_int)), ("cl_birth_from", C.POINTER(C.c_int)), ("cl_birth_ham", C.POINTER(C.c_int)),
        ("cl_center", C.POINTER(C.c_int)),
        ("cl_pval", C.POINTER(C.c_double)), ("cl_birth_pval", C.POINTER(C.c_double)),
        ("cl_birth_fold", C.POINTER(C.c_double)), ("cl_birth_qave", C.POINTER(C.c_double)),
        ("bs_pos", C.POINTER(C.c_int)), ("bs_clust", C.POINTER(C.c_int)),
        ("bs_ref", C.POINTER(C.c_char)), ("bs_sub", C.POINTER(C.c_char)), ("bs_qual", C.POINTER(C.c_double)),
        ("subqual", C.POINTER(C.c_int)), ("clusterquals", C.POINTER(C.c_double)),
        ("map", C.POINTER(C.c_int)), ("pval", C.POINTER(C.c_double)),
    ]


def build():
    """(Re)build libdada2oracle.so (and _ref where /root/reference exists) via oracle/Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        L = C.CDLL(_PATH)
        L.oracle_run.restype = C.POINTER(_CResult)
        L.oracle_run.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_int, C.POINTER(COpts), C.c_char_p, C.c_int]
        L.oracle_result_free.argtypes = [C.POINTER(_CResult)]
        L.oracle_nwalign.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                     C.c_char_p]
        L.oracle_compare.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.POINTER(COpts), C.c_double, C.c_void_p]
        L.oracle_calc_pA.restype = C.c_double
        L.oracle_calc_pA.argtypes = [C.c_int, C.c_double, C.c_int]
        L.dada2_oracle_ppois.restype = C.c_double
        L.dada2_oracle_ppois.argtypes = [C.c_double, C.c_double, C.c_int]
        _lib = L
    return _lib


def _arr(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dtype=dt)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)


def dada_uniques(seqs, abundances, priors, err, quals, opts: DadaOpts = None, *, max_clust=None,
                 multithread=False, verbose=False, copts: COpts = None) -> DadaResult:
    L = lib()
    co = copts if copts is not None else (opts or DadaOpts()).to_c(max_clust=max_clust, multithread=multithread,
                                                                  verbose=verbose)
    arr, ab, pr, ef, ncol, q, qn = pack_inputs(seqs, abundances, priors, err, quals)
    eb = C.create_string_buffer(1024)
    rp = L.oracle_run(len(seqs), arr, ab.ctypes.data, pr.ctypes.data, ef.ctypes.data, ncol,
                      q.ctypes.data if q is not None else None, qn, C.byref(co), eb, 1024)
    if not rp:
        raise RuntimeError(eb.value.decode())
    try:
        r = rp.contents
        Cn, nb = r.nclust, r.nbirth_subs
        clustering = {
            "sequence": [r.cl_sequence[i].decode() for i in range(Cn)],
            "abundance": _arr(r.cl_abundance, Cn, np.int32), "n0": _arr(r.cl_n0, Cn, np.int32),
            "n1": _arr(r.cl_n1, Cn, np.int32), "nunq": _arr(r.cl_nunq, Cn, np.int32),
            "pval": _arr(r.cl_pval, Cn, np.float64), "birth_from": _arr(r.cl_birth_from, Cn, np.int32),
            "birth_pval": _arr(r.cl_birth_pval, Cn, np.float64), "birth_fold": _arr(r.cl_birth_fold, Cn, np.float64),
            "birth_ham": _arr(r.cl_birth_ham, Cn, np.int32), "birth_qave": _arr(r.cl_birth_qave, Cn, np.float64),
        }
        birth_subs = {
            "pos": _arr(r.bs_pos, nb, np.int32), "ref": [r.bs_ref[i].decode() for i in range(nb)],
            "sub": [r.bs_sub[i].decode() for i in range(nb)], "qual": _arr(r.bs_qual, nb, np.float64),
            "clust": _arr(r.bs_clust, nb, np.int32),
        }
        subqual = _arr(r.subqual, 16 * r.ncol, np.int32).reshape(r.ncol, 16).T.copy()
        cq = _arr(r.clusterquals, r.maxlen * Cn, np.float64).reshape(Cn, r.maxlen).T.copy()
        if q is None:
            cq = np.zeros((r.maxlen, Cn))
        return DadaResult(clustering, birth_subs, subqual, cq, _arr(r.map, r.nraw, np.int32),
                          _arr(r.pval, r.nraw, np.float64),
                          stats={"nalign": r.nalign, "nshroud": r.nshroud,
                                 "center": _arr(r.cl_center, Cn, np.int32)})
    finally:
        L.oracle_result_free(rp)


def nwalign(s1, s2, match=5, mismatch=-4, gap=-8, band=16, gapless=False):
    n = len(s1) + len(s2) + 2
    o0, o1 = C.create_string_buffer(n), C.create_string_buffer(n)
    lib().oracle_nwalign(s1.encode(), s2.encode(), match, mismatch, gap, band, int(gapless), o0, o1)
    return o0.value.decode(), o1.value.decode()


def C_nwalign(s1, s2, match=5, mismatch=-4, gap_p=-8, homo_gap_p=None, band=-1, endsfree=True):
    """C_nwalign (evaluate.cpp:18-62): nwalign_endsfree / nwalign_endsfree_homo / global nwalign, by the restatement."""
    L = lib()
    L.oracle_nwalign2.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    n = len(s1) + len(s2) + 2
    o0, o1 = C.create_string_buffer(n), C.create_string_buffer(n)
    L.oracle_nwalign2(s1.encode(), s2.encode(), match, mismatch, gap_p, gap_p if homo_gap_p is None else homo_gap_p, band,
                      int(endsfree), o0, o1)
    return o0.value.decode(), o1.value.decode()


def compare(cseq, cq, rseq, rq, err, opts: DadaOpts = None, kdist_cutoff=None):
    o = opts or DadaOpts()
    co = o.to_c()
    e = np.ascontiguousarray(np.asarray(err, dtype=np.float64))
    cqa = np.ascontiguousarray(cq, dtype=np.float64)
    rqa = np.ascontiguousarray(rq, dtype=np.float64)
    out = np.zeros(4)
    lib().oracle_compare(cseq.encode(), cqa.ctypes.data, rseq.encode(), rqa.ctypes.data, e.ctypes.data, e.shape[1],
                         C.byref(co), float(o.KDIST_CUTOFF if kdist_cutoff is None else kdist_cutoff),
                         out.ctypes.data)
    return float(out[0]), int(out[1]), float(out[2]), float(out[3])


def calc_pA(reads, E, prior):
    return lib().oracle_calc_pA(int(reads), float(E), int(bool(prior)))


def ppois_upper(x, lam):
    return lib().dada2_oracle_ppois(float(x), float(lam), 0)


# ---- bimera identification (restatement of src/chimera.cpp) ---------------------------------------
def _bim_args(L):
    L.oracle_table_bimera2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_char_p), C.c_double, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.oracle_is_bimera.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]


def table_bimera2(mat, seqs, min_fold=1.5, min_abund=2, allow_one_off=False, min_one_off_par_dist=4, match=5, mismatch=-4,
                  gap_p=-8, max_shift=16):
    """C_table_bimera2: mat is [nsamples, nseqs]; returns (nflag[nseqs], nsam[nseqs])."""
    L = lib()
    _bim_args(L)
    m = np.asfortranarray(np.asarray(mat, dtype=np.int32))
    nrow, ncol = m.shape
    arr = (C.c_char_p * ncol)(*[s.encode() for s in seqs])
    nflag, nsam = np.zeros(ncol, dtype=np.int32), np.zeros(ncol, dtype=np.int32)
    L.oracle_table_bimera2(nrow, ncol, m.ctypes.data, arr, float(min_fold), int(min_abund), int(allow_one_off),
                           int(min_one_off_par_dist), match, mismatch, gap_p, int(max_shift), nflag.ctypes.data, nsam.ctypes.data)
    return nflag, nsam


def bimera_pairs(queries, parents, allow_one_off=False, match=5, mismatch=-4, gap_p=-8, max_shift=16):
    """get_lr / get_ham_endsfree (chimera.cpp:211-293) per (query, parent) pair: int32 [n, 5]."""
    L = lib()
    n = len(queries)
    qa = (C.c_char_p * max(n, 1))(*[s.encode() for s in queries])
    pa = (C.c_char_p * max(n, 1))(*[s.encode() for s in parents])
    out = np.zeros((n, 5), dtype=np.int32)
    L.oracle_bimera_pairs.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.oracle_bimera_pairs.restype = None
    L.oracle_bimera_pairs(n, qa, pa, int(allow_one_off), match, mismatch, gap_p, int(max_shift), out.ctypes.data)
    return out


def is_bimera(sq, pars, allow_one_off=False, min_one_off_par_dist=4, match=5, mismatch=-4, gap_p=-8, max_shift=16):
    L = lib()
    _bim_args(L)
    arr = (C.c_char_p * max(1, len(pars)))(*[s.encode() for s in pars])
    return bool(L.oracle_is_bimera(sq.encode(), len(pars), arr, int(allow_one_off), int(min_one_off_par_dist), match, mismatch,
                                   gap_p, int(max_shift)))


def eval_pair(a1, a2):
    """Restatement of C_eval_pair (evaluate.cpp:73) -> (match, mismatch, indel)."""
    out = (C.c_int * 3)()
    if lib().oracle_eval_pair(a1.encode(), a2.encode(), out):
        return None
    return int(out[0]), int(out[1]), int(out[2])


def pair_consensus(a1, a2, prefer, trim_overhang=False):
    """Restatement of C_pair_consensus (evaluate.cpp:124)."""
    o = C.create_string_buffer(len(a1) + 2)
    if lib().oracle_pair_consensus(a1.encode(), a2.encode(), int(prefer), int(trim_overhang), o):
        return None
    return o.value.decode()
