"""ctypes front-end to oracle/_ref/libdada2ref.so — the reference's own C++ compiled in
place from /root/reference/src (recipe: oracle/Makefile).  TEST INFRASTRUCTURE ONLY:
imported by tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from dada2_amd.opts import COpts, DadaOpts, DadaResult, CLUSTERING_COLS, BIRTH_SUBS_COLS

_HERE = os.path.dirname(os.path.abspath(__file__))
# two builds of the same sources: -O2 (R's default flags; the parity target) and -O3 -march=x86-64-v3 (bench baseline)
_PATHS = {"O2": os.path.join(_HERE, "_ref", "libdada2ref.so"), "O3": os.path.join(_HERE, "_ref", "libdada2ref_o3.so"),
          # NOT the reference: the Rcpp glue TU of INTEGRATION.md over libdada2hip.so (tests/glue), which hands back the same
          # Rcpp::List through the same flat entry points - read here so that both sides go through one marshalling
          "glue": os.path.join(os.path.dirname(_HERE), "tests", "glue", "libdada2glue.so")}
_PATH = _PATHS["O2"]
_libs = {}


def available(flavour: str = "O2") -> bool:
    return os.path.exists(_PATHS[flavour])


def lib(flavour: str = "O2"):
    if flavour not in _libs:
        if not available(flavour):
            raise RuntimeError(f"{_PATHS[flavour]} missing: run `make -C oracle` where /root/reference exists")
        L = C.CDLL(_PATHS[flavour])
        L.ref_dada_uniques.restype = C.c_void_p
        L.ref_dada_uniques.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int, C.POINTER(COpts), C.c_char_p, C.c_int]
        L.ref_result_free.argtypes = [C.c_void_p]
        L.ref_result_get.restype = C.c_long
        L.ref_result_get.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.ref_result_str.restype = C.c_char_p
        L.ref_result_str.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_long]
        if flavour != "glue":
            L.ref_nwalign.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                      C.c_char_p, C.c_char_p, C.c_int]
            L.ref_compare.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.POINTER(COpts), C.c_double, C.c_void_p, C.c_char_p, C.c_int]
            L.ref_calc_pA.restype = C.c_double
            L.ref_calc_pA.argtypes = [C.c_int, C.c_double, C.c_int]
            L.dada2_oracle_ppois.restype = C.c_double
            L.dada2_oracle_ppois.argtypes = [C.c_double, C.c_double, C.c_int]
            L.dada2_shim_set_threads.argtypes = [C.c_int]
        _libs[flavour] = L
    return _libs[flavour]


def set_threads(n: int):
    for f in ("O2", "O3"):
        if available(f):
            lib(f).dada2_shim_set_threads(int(n))


def _get(h, a, b=None, L=None):
    L = L or lib()
    kind, nr, nc, data = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
    n = L.ref_result_get(h, a.encode(), (b or "").encode(), C.byref(kind), C.byref(nr), C.byref(nc), C.byref(data))
    if n < 0:
        raise KeyError((a, b))
    k = kind.value
    if k == 2:
        return [L.ref_result_str(h, a.encode(), (b or "").encode(), i).decode() for i in range(n)]
    dt = np.int32 if k in (0, 3) else np.float64
    if n == 0:
        arr = np.zeros(0, dtype=dt)
    else:
        arr = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_int32 if dt == np.int32 else C.c_double)), shape=(n,)).copy()
    if k in (3, 4):
        arr = arr.reshape(nc.value, nr.value).T  # column-major -> [nr, nc]
    return arr


class _CharPP:
    """`char **` over one NUL-separated blob (a ctypes array of a million c_char_p takes seconds to build)."""

    def __init__(self, seqs):
        parts = [s.encode("ascii") for s in seqs]
        self.blob = b"\0".join(parts) + b"\0"
        n = len(parts)
        lens = np.fromiter((len(x) for x in parts), dtype=np.int64, count=n)
        off = np.zeros(n, dtype=np.int64)
        if n > 1:
            np.cumsum(lens[:-1] + 1, out=off[1:])
        self.ptrs = (off + np.frombuffer(self.blob, dtype=np.uint8).ctypes.data).astype(np.uint64)
        self._as_parameter_ = C.cast(C.c_void_p(self.ptrs.ctypes.data if n else None), C.POINTER(C.c_char_p))


def pack_inputs(seqs, abundances, priors, err, quals):
    """Common marshalling: quals is the R-side [N, maxlen] matrix (one row per unique, NaN
    padded) or None; the boundary wants it transposed, maxlen x N column-major
    (R/dada.R:337 `t(drpi$quals)`, Rmain.cpp:69,113) == the same C-contiguous [N, maxlen] buffer."""
    n = len(seqs)
    arr = _CharPP(seqs)
    ab = np.ascontiguousarray(abundances, dtype=np.int32)
    pr = np.ascontiguousarray(priors if priors is not None else np.zeros(n), dtype=np.uint8)
    e = np.asarray(err, dtype=np.float64)
    assert e.shape[0] == 16
    ef = np.ascontiguousarray(e.T)  # column-major 16 x Q
    if quals is None:
        q, qn = None, 0
    else:
        q = np.ascontiguousarray(quals, dtype=np.float64)
        assert q.shape[0] == n
        qn = q.shape[1]
    return arr, ab, pr, ef, e.shape[1], q, qn


def dada_uniques(seqs, abundances, priors, err, quals, opts: DadaOpts = None, *, max_clust=None,
                 multithread=False, verbose=False, copts: COpts = None, flavour: str = "O2", packed=None,
                 call_seconds: list = None) -> DadaResult:
    """Run the reference's dada_uniques (Rmain.cpp:30).  ``packed`` = a pack_inputs() tuple prepared beforehand;
    ``call_seconds`` (a list) receives the wall time of the C call alone (no Python marshalling)."""
    import time
    L = lib(flavour)
    co = copts if copts is not None else (opts or DadaOpts()).to_c(max_clust=max_clust, multithread=multithread,
                                                                  verbose=verbose)
    arr, ab, pr, ef, ncol, q, qn = packed if packed is not None else pack_inputs(seqs, abundances, priors, err, quals)
    eb = C.create_string_buffer(1024)
    t0 = time.perf_counter()
    h = L.ref_dada_uniques(len(seqs), arr, ab.ctypes.data, pr.ctypes.data, ef.ctypes.data, ncol,
                           q.ctypes.data if q is not None else None, qn, C.byref(co), eb, 1024)
    if call_seconds is not None:
        call_seconds.append(time.perf_counter() - t0)
    if not h:
        raise RuntimeError(eb.value.decode())
    try:
        clustering = {c: _get(h, "clustering", c, L) for c in CLUSTERING_COLS}
        birth_subs = {c: _get(h, "birth_subs", c, L) for c in BIRTH_SUBS_COLS}
        return DadaResult(clustering, birth_subs, _get(h, "subqual", None, L), _get(h, "clusterquals", None, L),
                          _get(h, "map", None, L), _get(h, "pval", None, L))
    finally:
        L.ref_result_free(h)


WHICH = {"endsfree": 0, "vectorized": 1, "gapless": 2, "global": 3}


def nwalign(s1, s2, match=5, mismatch=-4, gap=-8, band=16, which="vectorized"):
    L = lib()
    n = len(s1) + len(s2) + 2
    o0, o1, eb = C.create_string_buffer(n), C.create_string_buffer(n), C.create_string_buffer(512)
    rc = L.ref_nwalign(s1.encode(), s2.encode(), match, mismatch, gap, band, WHICH[which], o0, o1, eb, 512)
    if rc:
        raise RuntimeError(eb.value.decode())
    return o0.value.decode(), o1.value.decode()


def nwvec_raw(s1, s2, match=5, mismatch=-4, gap=-8, band=16, endsfree=True):
    """C_nwvec's call on one pair of strings (nwalign_vectorized.cpp:321-343): raw bytes in, any letters."""
    L = lib()
    n = len(s1) + len(s2) + 2
    o0, o1, eb = C.create_string_buffer(n), C.create_string_buffer(n), C.create_string_buffer(512)
    rc = L.ref_nwvec_raw(s1.encode(), s2.encode(), match, mismatch, gap, band, int(endsfree), o0, o1, eb, 512)
    if rc:
        raise RuntimeError(eb.value.decode())
    return o0.value.decode(), o1.value.decode()


def compare(cseq, cq, rseq, rq, err, opts: DadaOpts = None, kdist_cutoff=None):
    """(lambda, hamming|-1, kdist, kodist) for one centre/raw pair via sub_new + compute_lambda_ts."""
    L = lib()
    o = (opts or DadaOpts())
    co = o.to_c()
    e = np.ascontiguousarray(np.asarray(err, dtype=np.float64))  # row-major [16, Q] as cluster.cpp:166-170
    cqa = np.ascontiguousarray(cq, dtype=np.float64)
    rqa = np.ascontiguousarray(rq, dtype=np.float64)
    out = np.zeros(4)
    eb = C.create_string_buffer(512)
    rc = L.ref_compare(cseq.encode(), cqa.ctypes.data, rseq.encode(), rqa.ctypes.data, e.ctypes.data, e.shape[1],
                       C.byref(co), float(o.KDIST_CUTOFF if kdist_cutoff is None else kdist_cutoff),
                       out.ctypes.data, eb, 512)
    if rc:
        raise RuntimeError(eb.value.decode())
    return float(out[0]), int(out[1]), float(out[2]), float(out[3])


def calc_pA(reads, E, prior):
    return lib().ref_calc_pA(int(reads), float(E), int(bool(prior)))


def ppois_upper(x, lam):
    return lib().dada2_oracle_ppois(float(x), float(lam), 0)


# ---- bimera identification by the reference itself (src/chimera.cpp compiled in place) -----------
def table_bimera2(mat, seqs, min_fold=1.5, min_abund=2, allow_one_off=False, min_one_off_par_dist=4, match=5, mismatch=-4,
                  gap_p=-8, max_shift=16):
    L = lib()
    m = np.asfortranarray(np.asarray(mat, dtype=np.int32))
    nrow, ncol = m.shape
    arr = (C.c_char_p * ncol)(*[s.encode() for s in seqs])
    nflag, nsam = np.zeros(ncol, dtype=np.int32), np.zeros(ncol, dtype=np.int32)
    eb = C.create_string_buffer(512)
    L.ref_table_bimera2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_char_p), C.c_double, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    rc = L.ref_table_bimera2(nrow, ncol, m.ctypes.data, arr, float(min_fold), int(min_abund), int(allow_one_off),
                             int(min_one_off_par_dist), match, mismatch, gap_p, int(max_shift), nflag.ctypes.data,
                             nsam.ctypes.data, eb, 512)
    if rc:
        raise RuntimeError(eb.value.decode())
    return nflag, nsam


def bimera_pairs(queries, parents, allow_one_off=False, match=5, mismatch=-4, gap_p=-8, max_shift=16):
    """The reference's get_lr / get_ham_endsfree on its own nwalign_vectorized2 alignment (chimera.cpp:26-36): int32 [n, 5]."""
    L = lib()
    n = len(queries)
    qa = (C.c_char_p * max(n, 1))(*[s.encode() for s in queries])
    pa = (C.c_char_p * max(n, 1))(*[s.encode() for s in parents])
    out = np.zeros((n, 5), dtype=np.int32)
    L.ref_bimera_pairs.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    if L.ref_bimera_pairs(n, qa, pa, int(allow_one_off), match, mismatch, gap_p, int(max_shift), out.ctypes.data) != 0:
        raise RuntimeError("get_lr failed")
    return out


def is_bimera(sq, pars, allow_one_off=False, min_one_off_par_dist=4, match=5, mismatch=-4, gap_p=-8, max_shift=16):
    L = lib()
    L.ref_is_bimera.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    arr = (C.c_char_p * max(1, len(pars)))(*[s.encode() for s in pars])
    r = L.ref_is_bimera(sq.encode(), len(pars), arr, int(allow_one_off), int(min_one_off_par_dist), match, mismatch, gap_p,
                        int(max_shift))
    if r < 0:
        raise RuntimeError("C_is_bimera failed")
    return bool(r)


def C_nwalign(s1, s2, match=5, mismatch=-4, gap_p=-8, homo_gap_p=None, band=-1, endsfree=True):
    """The R-visible aligner itself (evaluate.cpp:18): what R's nwalign() calls."""
    L = lib()
    n = len(s1) + len(s2) + 2
    o0, o1, eb = C.create_string_buffer(n), C.create_string_buffer(n), C.create_string_buffer(512)
    rc = L.ref_C_nwalign(s1.encode(), s2.encode(), match, mismatch, gap_p, gap_p if homo_gap_p is None else homo_gap_p, band,
                         int(endsfree), o0, o1, eb, 512)
    if rc:
        raise RuntimeError(eb.value.decode())
    return o0.value.decode(), o1.value.decode()


def eval_pair(a1, a2):
    """C_eval_pair (evaluate.cpp:73) -> (match, mismatch, indel)."""
    out = (C.c_int * 3)()
    if lib().ref_eval_pair(a1.encode(), a2.encode(), out):
        return None
    return int(out[0]), int(out[1]), int(out[2])


def pair_consensus(a1, a2, prefer, trim_overhang=False):
    """C_pair_consensus (evaluate.cpp:124)."""
    o = C.create_string_buffer(len(a1) + 2)
    if lib().ref_pair_consensus(a1.encode(), a2.encode(), int(prefer), int(trim_overhang), o):
        return None
    return o.value.decode()
