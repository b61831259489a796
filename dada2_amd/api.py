"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

* ``dada_uniques(...)``  — the ``.Call`` of R/RcppExports.R:8-10 (src/Rmain.cpp:30): same
  argument meaning, same six outputs, errors raised with the reference's messages.
* ``Sample``             — the same call split into "make the uniques resident in HBM" and
  "run with this error matrix", which is what the selfConsist loop of R/dada.R:256-405 needs.
* ``nwalign`` / ``nwvec`` — R/misc.R:179 ``nwalign()`` -> C_nwalign / C_nwvec.
* ``dada(...)``          — the per-sample loop + selfConsist loop of R/dada.R:144-487, reduced
  to what the hot path needs (single process; ``dada2_amd.multi`` shards samples over GPUs).

R is not installed in this image, so this Python layer stands where R/dada.R
stands; INTEGRATION.md shows the Rcpp stub that binds the same C ABI from R.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .io import Derep, extend_err
from .opts import COpts, DadaOpts, DadaResult

_EB = 2048


class HostInput:
    """Host-side inputs of one ``dada_uniques`` call laid out as the C ABI takes them — what R already holds in its
    own vectors when it issues the ``.Call``: ``char**`` strings, int32 abundances, uint8 priors and the column-major
    ``maxlen x nraw`` double quality matrix (here: row-major [nraw, maxlen], the same bytes).  Built once; timing a
    call on a ``HostInput`` measures the boundary call itself, not Python's marshalling."""

    def __init__(self, seqs, abundances, priors, quals):
        n = len(seqs)
        self.n = n
        if n and isinstance(seqs[0], bytes):
            parts = seqs
        else:
            parts = [s.encode("ascii") for s in seqs]
        self._blob = b"\0".join(parts) + b"\0"
        lens = np.fromiter((len(x) for x in parts), dtype=np.int64, count=n)
        off = np.zeros(n, dtype=np.int64)
        if n > 1:
            np.cumsum(lens[:-1] + 1, out=off[1:])
        base = np.frombuffer(self._blob, dtype=np.uint8).ctypes.data
        self._ptrs = (off + base).astype(np.uint64)             # const char *const *
        self.seqs_p = self._ptrs.ctypes.data if n else None
        self.ab = np.ascontiguousarray(abundances, dtype=np.int32)
        self.pr = None if priors is None else np.ascontiguousarray(priors, dtype=np.uint8)
        self.q = None if quals is None else np.ascontiguousarray(quals, dtype=np.float64)
        if self.q is not None and self.q.ndim == 2 and self.q.shape[0] != n:
            raise ValueError("derep$quals matrices must have one row for each derep$unique sequence.")
        self.qn = 0 if self.q is None else self.q.shape[1]

    @classmethod
    def from_derep(cls, d: Derep, priors=None):
        return cls(d.seqs, d.abundances, priors, d.quals)

    @property
    def nbytes(self):
        return len(self._blob) + self.ab.nbytes + (0 if self.q is None else self.q.nbytes) + 8 * self.n


def _pack(seqs, abundances, priors, quals):
    h = seqs if isinstance(seqs, HostInput) else HostInput(seqs, abundances, priors, quals)
    return h


def _err_colmajor(err):
    e = np.asarray(err, dtype=np.float64)
    if e.ndim != 2 or e.shape[0] != 16:
        raise _lib.Dada2HipError(1, "Error matrix must have 16 rows.")
    return np.ascontiguousarray(e.T), e.shape[1]


def _collect(L, h) -> DadaResult:
    Cn = L.dada2hip_result_nclust(h)
    N = L.dada2hip_result_nraw(h)
    ml = L.dada2hip_result_maxlen(h)
    nc = L.dada2hip_result_ncol(h)
    nb = L.dada2hip_result_nbirth_subs(h)

    def arr(name, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(getattr(L, "dada2hip_result_" + name)(h), shape=(n,)).astype(dt, copy=True)

    clustering = {
        "sequence": [L.dada2hip_result_sequence(h, i).decode() for i in range(Cn)],
        "abundance": arr("abundance", Cn, np.int32), "n0": arr("n0", Cn, np.int32), "n1": arr("n1", Cn, np.int32),
        "nunq": arr("nunq", Cn, np.int32), "pval": arr("clust_pval", Cn, np.float64),
        "birth_from": arr("birth_from", Cn, np.int32), "birth_pval": arr("birth_pval", Cn, np.float64),
        "birth_fold": arr("birth_fold", Cn, np.float64), "birth_ham": arr("birth_ham", Cn, np.int32),
        "birth_qave": arr("birth_qave", Cn, np.float64),
    }
    ref = L.dada2hip_result_bs_ref(h)
    sub = L.dada2hip_result_bs_sub(h)
    birth_subs = {
        # (one block copy + one decode each: a per-element ctypes read cost 10 ms of the boundary call at 10^4 substitutions)
        "pos": arr("bs_pos", nb, np.int32), "ref": list(C.string_at(ref, nb).decode("ascii")) if nb else [],
        "sub": list(C.string_at(sub, nb).decode("ascii")) if nb else [], "qual": arr("bs_qual", nb, np.float64),
        "clust": arr("bs_clust", nb, np.int32),
    }
    subqual = arr("subqual", 16 * nc, np.int32).reshape(nc, 16).T.copy()
    cq = arr("clusterquals", ml * Cn, np.float64).reshape(Cn, ml).T.copy()
    st = _lib.CStats()
    L.dada2hip_result_stats(h, C.byref(st))
    stats = st.as_dict()
    stats["center"] = arr("center", Cn, np.int32)
    return DadaResult(clustering, birth_subs, subqual, cq, arr("map", N, np.int32), arr("pval", N, np.float64), stats)


def _copts(opts, max_clust, multithread, verbose, copts):
    if copts is not None:
        return copts
    return (opts or DadaOpts()).to_c(max_clust=max_clust, multithread=multithread, verbose=verbose)


def dada_uniques(seqs, abundances, priors, err, quals, opts: DadaOpts = None, *, max_clust=None, multithread=False,
                 verbose=False, copts: COpts = None, device: int = 0, log=None, should_abort=None) -> DadaResult:
    """One ``dada_uniques`` call (src/Rmain.cpp:30) on the GPU.  ``quals`` is the derep-side
    [N, maxlen] matrix (NaN past a short read's end); ``err`` is 16 x Q.  ``log(str)`` receives the ``verbose`` lines
    (Rprintf, src/Rmain.cpp:317-333), ``should_abort()`` is polled once per divisive round (Rcpp::checkUserInterrupt,
    src/Rmain.cpp:330): a true value ends the call with ``Dada2HipError(code=5)``."""
    L = _lib.lib()
    co = _copts(opts, max_clust, multithread, verbose, copts)
    hooks, keep = _lib.make_hooks(log, should_abort)
    hi = _pack(seqs, abundances, priors, quals)
    e, ncol = _err_colmajor(err)
    eb = C.create_string_buffer(_EB)
    h = C.c_void_p()
    rc = L.dada2hip_dada_uniques(hi.n, hi.seqs_p, hi.ab.ctypes.data, hi.pr.ctypes.data if hi.pr is not None else None,
                                 e.ctypes.data, ncol, hi.q.ctypes.data if hi.q is not None else None, hi.qn, C.byref(co),
                                 device, C.byref(hooks) if hooks is not None else None, C.byref(h), eb, _EB)
    if keep and keep["error"] is not None:
        if rc == 0:
            L.dada2hip_result_free(h)
        raise keep["error"]
    _lib.check(rc, eb)
    try:
        return _collect(L, h)
    finally:
        L.dada2hip_result_free(h)


def dada_uniques_multi(inputs, err, opts: DadaOpts = None, *, devices=(0,), copts: COpts = None):
    """``dada2hip_run_multi``: the per-sample loop of R/dada.R:266 over the GPUs of the node, one host thread per entry
    of ``devices`` inside the library.  ``inputs``: list of HostInput (or Derep).  Returns list[DadaResult]."""
    L = _lib.lib()
    co = _copts(opts, None, False, False, copts)
    his = [x if isinstance(x, HostInput) else HostInput.from_derep(x) for x in inputs]
    n = len(his)
    arr = (_lib.CSampleInput * max(n, 1))()
    for i, hi in enumerate(his):
        arr[i].nraw = hi.n
        arr[i].quals_nrow = hi.qn
        arr[i].seqs = C.cast(C.c_void_p(hi.seqs_p), C.POINTER(C.c_char_p))
        arr[i].abundances = hi.ab.ctypes.data
        arr[i].priors = hi.pr.ctypes.data if hi.pr is not None else None
        arr[i].quals = hi.q.ctypes.data if hi.q is not None else None
    e, ncol = _err_colmajor(err)
    devs = np.ascontiguousarray(devices, dtype=np.int32)
    outs = (C.c_void_p * max(n, 1))()
    eb = C.create_string_buffer(_EB)
    rc = L.dada2hip_run_multi(n, arr, e.ctypes.data, ncol, C.byref(co), devs.size, devs.ctypes.data, outs, eb, _EB)
    _lib.check(rc, eb)
    res = []
    try:
        for i in range(n):
            res.append(_collect(L, C.c_void_p(outs[i])))
    finally:
        for i in range(n):
            if outs[i]:
                L.dada2hip_result_free(C.c_void_p(outs[i]))
    return res


class NativeDerep:
    """derepFastq through the library's host-side dereplicator (dada2hip_derep_fastq; R/sequenceIO.R:45-124): the object
    stays in the library's memory; ``to_derep`` copies it into the numpy ``Derep`` mirror, ``Sample.from_native`` uploads it
    as a resident sample without a Python-side copy."""

    def __init__(self, path: str, n: int = 10**6, qual_offset: int = 0):
        L = _lib.lib()
        eb = C.create_string_buffer(_EB)
        self._h = C.c_void_p()
        _lib.check(L.dada2hip_derep_fastq(str(path).encode(), int(n), int(qual_offset), C.byref(self._h), eb, _EB), eb)
        self.nuniques = L.dada2hip_derep_nuniques(self._h)
        self.nreads = L.dada2hip_derep_nreads(self._h)
        self.maxlen = L.dada2hip_derep_maxlen(self._h)

    def to_derep(self) -> Derep:
        L = _lib.lib()
        n, ml = self.nuniques, self.maxlen
        sp = L.dada2hip_derep_seqs(self._h)
        seqs = [sp[i].decode("ascii") for i in range(n)]
        ab = np.ctypeslib.as_array(L.dada2hip_derep_abundances(self._h), (n,)).copy()
        q = np.ctypeslib.as_array(L.dada2hip_derep_quals(self._h), (n, ml)).copy()
        mp = np.ctypeslib.as_array(L.dada2hip_derep_map(self._h), (self.nreads,)).copy() if self.nreads else np.zeros(0, np.int32)
        mp[mp == np.iinfo(np.int32).min] = -1
        return Derep(seqs, ab, q, mp)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().dada2hip_derep_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def derep_fastq(path: str, n: int = 10**6, qual_offset: int = 0) -> Derep:
    """derepFastq(fl, n) (R/sequenceIO.R:45) -> Derep, through dada2hip_derep_fastq."""
    nd = NativeDerep(path, n, qual_offset)
    try:
        return nd.to_derep()
    finally:
        nd.close()


class Sample:
    """Uniques of one sample resident in HBM (dada2hip_sample_*): 2-bit reads, rounded
    qualities and k-mer records are uploaded/built once and reused by every ``run``."""

    def __init__(self, seqs, abundances, priors, quals, device: int = 0):
        L = _lib.lib()
        hi = _pack(seqs, abundances, priors, quals)
        eb = C.create_string_buffer(_EB)
        self._h = C.c_void_p()
        rc = L.dada2hip_sample_create(hi.n, hi.seqs_p, hi.ab.ctypes.data, hi.pr.ctypes.data if hi.pr is not None else None,
                                      hi.q.ctypes.data if hi.q is not None else None, hi.qn, device, C.byref(self._h), eb, _EB)
        _lib.check(rc, eb)
        self.nraw = hi.n
        self.device = device

    @classmethod
    def from_derep(cls, d: Derep, priors=None, device: int = 0):
        return cls(d.seqs, d.abundances, priors, d.quals, device)

    @classmethod
    def from_native(cls, nd: "NativeDerep", priors=None, device: int = 0):
        """dada2hip_sample_from_derep: the library's derep object -> resident sample, no host copy in between."""
        self = cls.__new__(cls)
        eb = C.create_string_buffer(_EB)
        self._h = C.c_void_p()
        pr = None if priors is None else np.ascontiguousarray(priors, dtype=np.uint8)
        rc = _lib.lib().dada2hip_sample_from_derep(nd._h, pr.ctypes.data if pr is not None else None, device, C.byref(self._h), eb, _EB)
        _lib.check(rc, eb)
        self.nraw = nd.nuniques
        self.device = device
        return self

    def set_priors(self, priors):
        eb = C.create_string_buffer(_EB)
        pr = np.ascontiguousarray(priors, dtype=np.uint8)
        _lib.check(_lib.lib().dada2hip_sample_set_priors(self._h, pr.ctypes.data, eb, _EB), eb)

    def run(self, err, opts: DadaOpts = None, *, max_clust=None, multithread=False, verbose=False,
            copts: COpts = None, log=None, should_abort=None) -> DadaResult:
        L = _lib.lib()
        co = _copts(opts, max_clust, multithread, verbose, copts)
        e, ncol = _err_colmajor(err)
        eb = C.create_string_buffer(_EB)
        h = C.c_void_p()
        hooks, keep = _lib.make_hooks(log, should_abort)
        rc = L.dada2hip_sample_run(self._h, e.ctypes.data, ncol, C.byref(co), C.byref(hooks) if hooks is not None else None,
                                   C.byref(h), eb, _EB)
        if keep and keep["error"] is not None:
            if rc == 0:
                L.dada2hip_result_free(h)
            raise keep["error"]
        _lib.check(rc, eb)
        try:
            return _collect(L, h)
        finally:
            L.dada2hip_result_free(h)

    def run_sharded(self, err, opts: DadaOpts, rank: int, world: int, exchange, *, max_clust=None, copts: COpts = None) -> DadaResult:
        """``dada2hip_sample_run_sharded``: this process does the per-unique work of block ``rank`` of ``world``.  ``exchange(kind,
        send: bytes-like memoryview, recv: writable memoryview)`` is the collective of include/dada2hip.h's dada2hip_shard
        (kind 0 = all-gather of equal-size payloads in rank order, kind 1 = in-place all-reduce(sum) of int64);
        ``dada2_amd.shard`` provides it over torch.distributed.  Every rank gets the complete result."""
        L = _lib.lib()
        co = _copts(opts, max_clust, False, False, copts)
        e, ncol = _err_colmajor(err)
        failure = []

        def _cb(user, kind, send, nbytes, recv):
            try:
                n = int(nbytes)
                sv = (C.c_char * n).from_address(send) if n else (C.c_char * 0)()
                rv = (C.c_char * (n * world if kind == 0 else n)).from_address(recv) if n else (C.c_char * 0)()
                exchange(int(kind), memoryview(sv).cast("B"), memoryview(rv).cast("B"))
                return 0
            except BaseException as ex:   # never unwind through the C frames
                failure.append(ex)
                return 1
        cb = _lib.EXCHANGE_FN(_cb)
        sh = _lib.CShard(int(rank), int(world), cb, None)
        eb = C.create_string_buffer(_EB)
        h = C.c_void_p()
        rc = L.dada2hip_sample_run_sharded(self._h, e.ctypes.data, ncol, C.byref(co), None, C.byref(sh), C.byref(h), eb, _EB)
        if failure:
            raise failure[0]
        _lib.check(rc, eb)
        try:
            return _collect(L, h)
        finally:
            L.dada2hip_result_free(h)

    def compare(self, centre: int, err, opts: DadaOpts = None, kdist_cutoff=None, skip=None):
        """One b_compare round (cluster.cpp:90-149): (lambda[N], hamming[N], cls[N], stats)."""
        L = _lib.lib()
        o = opts or DadaOpts()
        co = o.to_c()
        e, ncol = _err_colmajor(err)
        lam = np.zeros(self.nraw)
        ham = np.zeros(self.nraw, dtype=np.uint32)
        cls = np.zeros(self.nraw, dtype=np.uint8)
        sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
        st = _lib.CStats()
        eb = C.create_string_buffer(_EB)
        rc = L.dada2hip_sample_compare(self._h, int(centre), e.ctypes.data, ncol, C.byref(co),
                                       float(o.KDIST_CUTOFF if kdist_cutoff is None else kdist_cutoff),
                                       sk.ctypes.data if sk is not None else None, lam.ctypes.data, ham.ctypes.data,
                                       cls.ctypes.data, C.byref(st), eb, _EB)
        _lib.check(rc, eb)
        return lam, ham, cls, st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().dada2hip_sample_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def nwvec(s1, s2, match=5, mismatch=-4, gap=-8, band=-1, endsfree=True, device: int = 0):
    """C_nwvec (src/nwalign_vectorized.cpp:321): list of (al0, al1) for the pairs."""
    L = _lib.lib()
    n = len(s1)
    if n != len(s2):
        raise ValueError("Character vectors to be aligned must be of equal length.")
    a = (C.c_char_p * n)(*[x.encode() for x in s1])
    b = (C.c_char_p * n)(*[x.encode() for x in s2])
    bufs = [C.create_string_buffer(len(s1[i // 2]) + len(s2[i // 2]) + 2) for i in range(2 * n)]
    out = (C.c_char_p * (2 * n))(*[C.cast(x, C.c_char_p) for x in bufs])
    eb = C.create_string_buffer(_EB)
    _lib.check(L.dada2hip_nwvec(n, a, b, match, mismatch, gap, band, int(endsfree), device, out, eb, _EB), eb)
    return [(bufs[2 * i].value.decode(), bufs[2 * i + 1].value.decode()) for i in range(n)]


def nwalign(s1, s2, match=5, mismatch=-4, gap=-8, homo_gap=None, band=-1, endsfree=True, device: int = 0):
    """R/misc.R:179 nwalign(): one pair, returns (al0, al1)."""
    L = _lib.lib()
    o0 = C.create_string_buffer(len(s1) + len(s2) + 2)
    o1 = C.create_string_buffer(len(s1) + len(s2) + 2)
    eb = C.create_string_buffer(_EB)
    rc = L.dada2hip_nwalign(s1.encode(), s2.encode(), match, mismatch, gap, gap if homo_gap is None else homo_gap, band,
                            int(endsfree), device, o0, o1, eb, _EB)
    _lib.check(rc, eb)
    return o0.value.decode(), o1.value.decode()


def table_bimera2(mat, seqs, min_fold=1.5, min_abund=2, allow_one_off=False, min_one_off_par_dist=4, match=5, mismatch=-4,
                  gap_p=-8, max_shift=16, device: int = 0):
    """C_table_bimera2 (src/chimera.cpp:192; called by isBimeraDenovoTable, R/chimeras.R:236): ``mat`` is the
    [samples, sequences] count table; returns (nflag[nseq], nsam[nseq])."""
    L = _lib.lib()
    m = np.asfortranarray(np.asarray(mat, dtype=np.int32))
    nrow, ncol = m.shape
    if ncol != len(seqs):
        raise ValueError("The sequence table must have one column per sequence.")
    arr = (C.c_char_p * max(ncol, 1))(*[s.encode("ascii") for s in seqs])
    nflag, nsam = np.zeros(ncol, dtype=np.int32), np.zeros(ncol, dtype=np.int32)
    eb = C.create_string_buffer(_EB)
    _lib.check(L.dada2hip_table_bimera2(nrow, ncol, m.ctypes.data, arr, float(min_fold), int(min_abund), int(allow_one_off),
                                        int(min_one_off_par_dist), match, mismatch, gap_p, int(max_shift), device,
                                        nflag.ctypes.data, nsam.ctypes.data, eb, _EB), eb)
    return nflag, nsam


def is_bimera(sq, parents, allow_one_off=False, min_one_off_par_dist=4, match=5, mismatch=-4, gap_p=-8, max_shift=16, device: int = 0):
    """C_is_bimera (src/chimera.cpp:18; R/chimeras.R:43 isBimera)."""
    L = _lib.lib()
    arr = (C.c_char_p * max(len(parents), 1))(*[s.encode("ascii") for s in parents])
    out = C.c_int32(0)
    eb = C.create_string_buffer(_EB)
    _lib.check(L.dada2hip_is_bimera(sq.encode("ascii"), len(parents), arr, int(allow_one_off), int(min_one_off_par_dist), match,
                                    mismatch, gap_p, int(max_shift), device, C.byref(out), eb, _EB), eb)
    return bool(out.value)


def bimera_pairs(queries, parents, allow_one_off=False, match=5, mismatch=-4, gap_p=-8, max_shift=16, device: int = 0):
    """get_lr / get_ham_endsfree (src/chimera.cpp:211-293) of each (query, parent) alignment: int32 [n, 5] =
    left, right, left_oo, right_oo, hamming - what C_is_bimera / C_table_bimera2 reduce to a flag."""
    L = _lib.lib()
    n = len(queries)
    if n != len(parents):
        raise ValueError("queries and parents must have the same length")
    qa = (C.c_char_p * max(n, 1))(*[s.encode("ascii") for s in queries])
    pa = (C.c_char_p * max(n, 1))(*[s.encode("ascii") for s in parents])
    out = np.zeros((n, 5), dtype=np.int32)
    eb = C.create_string_buffer(_EB)
    _lib.check(L.dada2hip_bimera_pairs(n, qa, pa, int(allow_one_off), match, mismatch, gap_p, int(max_shift), device,
                                       out.ctypes.data, eb, _EB), eb)
    return out


def is_bimera_denovo_table(mat, seqs, min_sample_fraction=0.9, ignore_n_negatives=1, **kw):
    """isBimeraDenovoTable (R/chimeras.R:220-247): the consensus decision over samples on top of C_table_bimera2."""
    nflag, nsam = table_bimera2(mat, seqs, **kw)
    return (nflag >= nsam) | ((nflag > 0) & (nflag >= (nsam - ignore_n_negatives) * min_sample_fraction))


def merge_pairs(dadaF: DadaResult, derepF: Derep, dadaR: DadaResult, derepR: Derep, min_overlap=12, max_mismatch=0,
                return_rejects=False, just_concatenate=False, trim_overhang=False, device: int = 0):
    """mergePairs() for one sample (R/paired.R:92-201) through dada2hip_merge_pairs: a list of dict rows (sequence,
    abundance, forward, reverse, nmatch, nmismatch, nindel, prefer, accept) in the reference's order."""
    L = _lib.lib()
    NA = np.iinfo(np.int32).min
    mF, mR = np.asarray(derepF.map, dtype=np.int64), np.asarray(derepR.map, dtype=np.int64)
    # paired.R:115-119: the maps must be as long as each other and their largest entry (1-based there, 0-based here) must
    # name the LAST unique of the dada-class object - equality, not just "in range"
    def _max_ok(m, n):
        ok = m[m >= 0]
        return ok.size > 0 and int(ok.max()) + 1 == n
    if len(mF) != len(mR) or not _max_ok(mF, len(dadaF.map)) or not _max_ok(mR, len(dadaR.map)):
        raise _lib.Dada2HipError(1, "Non-corresponding derep-class and dada-class objects.")
    def denoised(dmap, rmap):
        dm = np.asarray(dmap, dtype=np.int64)
        out = np.full(len(rmap), NA, dtype=np.int32)
        ok = rmap >= 0
        v = dm[rmap[ok]]
        out[ok] = np.where(v > 0, v, NA)
        return out
    fwd, rev = denoised(dadaF.map, mF), denoised(dadaR.map, mR)
    sF, sR = list(dadaF.clustering["sequence"]), list(dadaR.clustering["sequence"])
    aF = (C.c_char_p * max(1, len(sF)))(*[x.encode() for x in sF])
    aR = (C.c_char_p * max(1, len(sR)))(*[x.encode() for x in sR])
    n0F = np.ascontiguousarray(dadaF.clustering["n0"], dtype=np.int32)
    n0R = np.ascontiguousarray(dadaR.clustering["n0"], dtype=np.int32)
    eb = C.create_string_buffer(_EB)
    h = C.c_void_p()
    _lib.check(L.dada2hip_merge_pairs(len(fwd), fwd.ctypes.data, rev.ctypes.data, len(sF), aF, n0F.ctypes.data, len(sR), aR,
                                      n0R.ctypes.data, int(min_overlap), int(max_mismatch), int(trim_overhang),
                                      int(just_concatenate), device, C.byref(h), eb, _EB), eb)
    try:
        n = L.dada2hip_mergers_nrow(h)
        col = {k: np.ctypeslib.as_array(getattr(L, "dada2hip_mergers_" + k)(h), (n,)).copy() if n else np.zeros(0, np.int32)
               for k in ("abundance", "forward", "reverse", "nmatch", "nmismatch", "nindel", "prefer", "accept")}
        rows = []
        for i in range(n):
            rows.append({"sequence": L.dada2hip_mergers_sequence(h, i).decode("ascii"), "abundance": int(col["abundance"][i]),
                         "forward": int(col["forward"][i]), "reverse": int(col["reverse"][i]), "nmatch": int(col["nmatch"][i]),
                         "nmismatch": int(col["nmismatch"][i]), "nindel": int(col["nindel"][i]),
                         "prefer": None if col["prefer"][i] == NA else int(col["prefer"][i]), "accept": bool(col["accept"][i])})
    finally:
        L.dada2hip_mergers_free(h)
    return rows if return_rejects else [r for r in rows if r["accept"]]


def calc_pA_device(reads, E, prior, device: int = 0):
    """calc_pA (src/pval.cpp:44-64) evaluated by the device kernel."""
    L = _lib.lib()
    r = np.ascontiguousarray(reads, dtype=np.int32)
    e = np.ascontiguousarray(E, dtype=np.float64)
    p = np.ascontiguousarray(prior, dtype=np.uint8)
    out = np.zeros(r.size)
    eb = C.create_string_buffer(_EB)
    _lib.check(L.dada2hip_calc_pA(r.size, r.ctypes.data, e.ctypes.data, p.ctypes.data, device, out.ctypes.data, eb, _EB), eb)
    return out


# --------------------------------------------------------------------------------------------------
def accumulate_trans(trans_list):
    """R/errorModels.R:462-471."""
    maxcol = max(t.shape[1] for t in trans_list)
    out = np.zeros((16, maxcol), dtype=np.int64)
    for t in trans_list:
        out[:, : t.shape[1]] += t
    return out


def noqual_errfun(trans, pseudocount=1):
    """R/errorModels.R:222-249 (noqualErrfun): one rate per transition, aggregated over quality."""
    trans = np.asarray(trans, dtype=np.float64)
    obs = trans.sum(axis=1) + pseudocount
    err = np.zeros_like(trans)
    for i in range(4):
        tot = obs[4 * i: 4 * i + 4].sum()
        rates = obs[4 * i: 4 * i + 4] / tot
        for j in range(4):
            if i != j:
                err[4 * i + j, :] = rates[j]
        err[5 * i, :] = 1.0 - sum(rates[j] for j in range(4) if j != i)
    return err


def dada(dereps, err=None, *, self_consist=False, err_fun=noqual_errfun, opts: DadaOpts = None, priors=None,
         device: int = 0, verbose=False, samples=None, timings: list = None, host_input=None, on_pass=None):
    """The sample loop and selfConsist loop of R/dada.R:256-405 over resident samples.

    ``err_fun`` maps the accumulated 16 x Q transition counts to a new error matrix; the
    reference's default is ``loessErrfun`` (stats::loess, third-party, not available here) so the
    deterministic ``noqualErrfun`` stands in.  Returns (list[DadaResult], err_out,
    list of err matrices tried).  ``on_pass(k, errs_used, max_clust, results)`` is called after every pass with the
    error matrix each sample was run with (the all-ones start of R/dada.R:298 included) - the parity tests check every
    pass of the loop through it, not just the last."""
    o = (opts or DadaOpts()).normalised()
    single = isinstance(dereps, Derep)
    if single:
        dereps = [dereps]
    import time
    own = samples is None
    t_create = time.perf_counter()
    if own:
        if host_input is not None and single:          # inputs already marshalled (bench.py): one resident sample from them
            samples = [Sample(host_input, None, None, None, device)]
        else:
            samples = [Sample.from_derep(d, None if priors is None else priors[i], device) for i, d in enumerate(dereps)]
    if timings is not None:
        timings.append((time.perf_counter() - t_create) * 1e3)   # [0] = making the samples resident (upload), then one entry per pass
    initialize = self_consist and err is None
    nconsist = 0 if initialize else 1
    errs = []
    # the largest rounded quality of each sample (R/dada.R:297-313 extends err to it): a property of the sample, taken ONCE - as a
    # statement inside the pass loop it re-read the whole maxlen x nraw double matrix every pass (2 GB at 10^6 uniques: 70 ms of
    # every selfConsist pass, profiles/r09a_selfconsist_passes.txt)
    qmaxes = [d.qmax() for d in dereps]
    try:
        while True:
            if nconsist > 0:
                errs.append(np.array(err, copy=True))
            results, used = [], []
            t_pass = time.perf_counter()
            for d, smp, qmax in zip(dereps, samples, qmaxes):
                erri = np.ones((16, max(41, qmax + 1))) if initialize else extend_err(err, qmax)   # R/dada.R:297-313
                results.append(smp.run(erri, o, max_clust=1 if initialize else None, verbose=verbose))
                used.append(erri)
            if timings is not None:
                timings.append((time.perf_counter() - t_pass) * 1e3)
            if on_pass is not None:
                on_pass(len(errs) if not initialize else 0, used, 1 if initialize else None, results)
            cur = accumulate_trans([r.subqual for r in results])
            new_err = err_fun(cur) if err_fun is not None else None
            if initialize:
                initialize = False
                new_err[[0, 5, 10, 15], :] = 1.0                                                   # R/dada.R:385-388
            err = new_err
            if (not self_consist) or any(np.array_equal(e, err) for e in errs) or nconsist >= o.MAX_CONSIST:
                break
            nconsist += 1
    finally:
        if own:
            for smp in samples:
                smp.close()
    return (results[0] if single else results), err, errs
