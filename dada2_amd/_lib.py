"""ctypes binding of libdada2hip.so (include/dada2hip.h).  There is deliberately no
fallback: if the HIP library has not been built, or no GPU is usable, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

from .opts import COpts

# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) when it initialises; a run uses three
# streams and several samples in flight three each (csrc/knobs.h, knobs_process_defaults: the library sets the same default when
# it is loaded - this line covers a torch that initialises the runtime between the import of this package and the first call).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdada2hip.so")
_lib = None


class Dada2HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class CStats(C.Structure):
    _fields_ = [
        ("ncompare", C.c_uint64), ("nskipped", C.c_uint64), ("nshroud", C.c_uint64), ("ngapless", C.c_uint64),
        ("nnw", C.c_uint64), ("nshuffle", C.c_uint64), ("nstored", C.c_uint64),
        ("rounds", C.c_uint32), ("kernel_times_sampled", C.c_uint32),
        ("ms_total", C.c_double), ("ms_upload", C.c_double), ("ms_screen", C.c_double), ("ms_nw", C.c_double),
        ("ms_gapless", C.c_double), ("ms_bookkeep", C.c_double), ("ms_pval", C.c_double), ("ms_final", C.c_double),
        ("nw_kernel_ms", C.c_double), ("nw_kernel_launches", C.c_uint64), ("nw_cells", C.c_uint64),
        ("screen_kernel_ms", C.c_double), ("screen_kernel_launches", C.c_uint64), ("screen_bytes", C.c_uint64),
        ("dev_ms_screen", C.c_double), ("dev_ms_nw", C.c_double), ("dev_ms_shuffle", C.c_double),
        ("dev_ms_pval", C.c_double), ("dev_ms_birth", C.c_double), ("dev_ms_final", C.c_double),
        ("ms_wait_device", C.c_double), ("ms_replay", C.c_double), ("ms_enqueue", C.c_double),
        ("nmoves", C.c_uint64), ("batch_compares", C.c_uint64),
        ("nnw_run", C.c_uint64), ("ngapless_run", C.c_uint64),
        ("lite_chains", C.c_uint64), ("lite_misses", C.c_uint64),
        ("tail_launches", C.c_uint64), ("tail_pauses", C.c_uint64), ("tail_levels", C.c_uint64),
        ("tail_blocks", C.c_uint32), ("tail_fallbacks", C.c_uint32),
        ("dev_ms_tail", C.c_double), ("tail_ms_entry", C.c_double), ("tail_ms_shuffle0", C.c_double),
        ("tail_ms_shuffle_more", C.c_double), ("tail_ms_pupdate", C.c_double), ("tail_ms_barriers", C.c_double),
        ("tail_ms_birth", C.c_double), ("tail_ms_publish", C.c_double), ("tail_ms_release", C.c_double),
        ("pf_compares", C.c_uint64), ("pf_hits", C.c_uint64), ("pf_waits", C.c_uint64), ("pf_exits", C.c_uint64),
        ("pf_centres", C.c_uint64), ("tail_threads", C.c_uint32), ("overlap_on", C.c_uint32),
        ("dev_ms_pf_screen", C.c_double), ("dev_ms_pf_nw", C.c_double),
        ("tail_xcd_barrier", C.c_uint32), ("tail_mirror", C.c_uint32),
        ("ms_setup", C.c_double), ("ms_round0", C.c_double), ("tail_ms_pf_wait", C.c_double), ("tail_ms_pf_plan", C.c_double),
        ("nnw_retry", C.c_uint64), ("nnw_fast", C.c_uint64), ("screen_stage2", C.c_uint64), ("nnw_rounds", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


# dada2hip_shard (include/dada2hip.h): rank / world + the collective the library calls at its exchange points
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p)


# dada2hip_hooks (include/dada2hip.h): the verbose log lines (Rprintf, src/Rmain.cpp:317-333) and the abort poll
# (Rcpp::checkUserInterrupt, src/Rmain.cpp:330)
LOG_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
ABORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class CHooks(C.Structure):
    _fields_ = [("log", LOG_FN), ("should_abort", ABORT_FN), ("user", C.c_void_p)]


def make_hooks(log=None, should_abort=None):
    """(CHooks, keepalive) for ``log(str)`` / ``should_abort() -> bool`` callables, or (None, None) when both are None.
    An exception raised by a callback cannot cross the C frame: it is recorded in ``keepalive['error']`` (and aborts the run
    when it comes from ``should_abort``); the caller re-raises it after the library call."""
    if log is None and should_abort is None:
        return None, None
    keep = {"error": None}

    def _log(msg, _user):
        try:
            if log is not None:
                log(msg.decode(errors="replace"))
        except BaseException as ex:   # noqa: BLE001
            keep["error"] = keep["error"] or ex

    def _abort(_user):
        try:
            return 1 if (should_abort is not None and should_abort()) else 0
        except BaseException as ex:   # noqa: BLE001
            keep["error"] = keep["error"] or ex
            return 1

    h = CHooks(LOG_FN(_log), ABORT_FN(_abort), None)
    keep["cb"] = (h.log, h.should_abort)
    return h, keep


class CShard(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("exchange", EXCHANGE_FN), ("user", C.c_void_p)]


# every symbol include/dada2hip.h declares (tests/test_abi.py checks the .so exports all of them)
EXPORTS = [
    "dada2hip_dada_uniques", "dada2hip_sample_create", "dada2hip_sample_set_priors", "dada2hip_sample_run",
    "dada2hip_sample_free", "dada2hip_sample_run_sharded", "dada2hip_sample_nraw", "dada2hip_sample_maxlen", "dada2hip_result_nclust",
    "dada2hip_result_nraw", "dada2hip_result_maxlen", "dada2hip_result_ncol", "dada2hip_result_nbirth_subs",
    "dada2hip_result_sequence", "dada2hip_result_abundance", "dada2hip_result_n0", "dada2hip_result_n1",
    "dada2hip_result_nunq", "dada2hip_result_clust_pval", "dada2hip_result_birth_from", "dada2hip_result_birth_pval",
    "dada2hip_result_birth_fold", "dada2hip_result_birth_ham", "dada2hip_result_birth_qave", "dada2hip_result_center",
    "dada2hip_result_bs_pos", "dada2hip_result_bs_ref", "dada2hip_result_bs_sub", "dada2hip_result_bs_qual",
    "dada2hip_result_bs_clust", "dada2hip_result_subqual", "dada2hip_result_clusterquals", "dada2hip_result_map",
    "dada2hip_result_pval", "dada2hip_result_stats", "dada2hip_result_free", "dada2hip_nwalign", "dada2hip_nwvec",
    "dada2hip_sample_compare", "dada2hip_calc_pA", "dada2hip_version", "dada2hip_run_multi", "dada2hip_trim_cache",
    "dada2hip_table_bimera2", "dada2hip_is_bimera", "dada2hip_bimera_pairs", "dada2hip_derep_fastq", "dada2hip_derep_nuniques",
    "dada2hip_derep_nreads", "dada2hip_derep_maxlen", "dada2hip_derep_seqs", "dada2hip_derep_abundances",
    "dada2hip_derep_quals", "dada2hip_derep_map", "dada2hip_derep_free", "dada2hip_sample_from_derep",
    "dada2hip_merge_pairs", "dada2hip_mergers_nrow", "dada2hip_mergers_sequence", "dada2hip_mergers_abundance",
    "dada2hip_mergers_forward", "dada2hip_mergers_reverse", "dada2hip_mergers_nmatch", "dada2hip_mergers_nmismatch",
    "dada2hip_mergers_nindel", "dada2hip_mergers_prefer", "dada2hip_mergers_accept", "dada2hip_mergers_free",
]


class CSampleInput(C.Structure):
    """dada2hip_sample_input"""
    _fields_ = [("nraw", C.c_int32), ("quals_nrow", C.c_int32), ("seqs", C.POINTER(C.c_char_p)),
                ("abundances", C.c_void_p), ("priors", C.c_void_p), ("quals", C.c_void_p)]


HIP_RUNTIME = {"chosen": "system", "path": None, "note": "no torch runtime looked for yet"}   # which libamdhip64 serves this process


def _soname_major(path):
    """Major version in the DT_SONAME of a libamdhip64 file (``libamdhip64.so.7`` -> 7), read from the file's string table
    without loading it; None if it cannot be told."""
    import mmap
    import re
    try:
        with open(path, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as m:
            hit = re.search(rb"libamdhip64\.so\.(\d+)\x00", m)
            return int(hit.group(1)) if hit else None
    except Exception:
        return None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  A ROCm build of torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): whichever
    copy is mapped first serves both torch and this library, and torch does not see a device through the system copy.  So if
    torch is INSTALLED (it is not imported here, and nothing of torch is used) its copy is mapped before libdada2hip.so asks
    for the SONAME - `import dada2_amd` and `import torch` then work in either order.  Only a copy of the SAME SONAME major as
    the runtime the library was built against is taken (a wheel from another ROCm generation must not silently serve
    the library's HIP calls); the choice is recorded in ``HIP_RUNTIME`` and reported by ``runtime_info()``.
    DADA2HIP_SYSTEM_HIP=1 skips this."""
    if os.environ.get("DADA2HIP_SYSTEM_HIP"):
        HIP_RUNTIME.update(chosen="system", note="DADA2HIP_SYSTEM_HIP is set")
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
        if not (cand and os.path.exists(cand)):
            HIP_RUNTIME.update(chosen="system", note="no torch-bundled libamdhip64 found")
            return
        built = None
        for sysdir in ("/opt/rocm/lib", "/opt/rocm/lib64"):
            if os.path.exists(os.path.join(sysdir, "libamdhip64.so")):
                built = _soname_major(os.path.realpath(os.path.join(sysdir, "libamdhip64.so")))
                break
        theirs = _soname_major(cand)
        if built is not None and theirs is not None and built != theirs:
            HIP_RUNTIME.update(chosen="system", path=None,
                               note=f"torch bundles libamdhip64.so.{theirs}, the library was built against .so.{built}: not shared")
            return
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        HIP_RUNTIME.update(chosen="torch", path=cand, note=f"SONAME major {theirs} (system: {built})")
    except Exception as ex:   # no torch, or its runtime does not load here: the system copy serves
        HIP_RUNTIME.update(chosen="system", note=f"torch runtime not mapped: {type(ex).__name__}")


def runtime_info():
    """Version string of the library and which HIP runtime serves it in this process."""
    L = lib()
    return {"library": L.dada2hip_version().decode(), "hip_runtime": dict(HIP_RUNTIME)}


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Dada2HipError(2, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, ip, cp = C.c_void_p, C.c_int32, C.c_char_p
    L.dada2hip_version.restype = cp
    L.dada2hip_sample_create.argtypes = [ip, vp, vp, vp, vp, ip, ip, C.POINTER(vp), cp, C.c_size_t]
    L.dada2hip_sample_set_priors.argtypes = [vp, vp, cp, C.c_size_t]
    L.dada2hip_sample_run.argtypes = [vp, vp, ip, C.POINTER(COpts), vp, C.POINTER(vp), cp, C.c_size_t]
    L.dada2hip_sample_run_sharded.argtypes = [vp, vp, ip, C.POINTER(COpts), vp, C.POINTER(CShard), C.POINTER(vp), cp, C.c_size_t]
    L.dada2hip_sample_free.argtypes = [vp]
    L.dada2hip_sample_nraw.argtypes = [vp]
    L.dada2hip_sample_maxlen.argtypes = [vp]
    L.dada2hip_dada_uniques.argtypes = [ip, vp, vp, vp, vp, ip, vp, ip, C.POINTER(COpts), ip, vp,
                                        C.POINTER(vp), cp, C.c_size_t]
    for name in ("nclust", "nraw", "maxlen", "ncol", "nbirth_subs"):
        getattr(L, "dada2hip_result_" + name).argtypes = [vp]
        getattr(L, "dada2hip_result_" + name).restype = ip
    L.dada2hip_result_sequence.argtypes = [vp, ip]
    L.dada2hip_result_sequence.restype = cp
    for name in ("abundance", "n0", "n1", "nunq", "birth_from", "birth_ham", "center", "bs_pos", "bs_clust", "subqual",
                 "map"):
        f = getattr(L, "dada2hip_result_" + name)
        f.argtypes = [vp]
        f.restype = C.POINTER(C.c_int32)
    for name in ("clust_pval", "birth_pval", "birth_fold", "birth_qave", "bs_qual", "clusterquals", "pval"):
        f = getattr(L, "dada2hip_result_" + name)
        f.argtypes = [vp]
        f.restype = C.POINTER(C.c_double)
    for name in ("bs_ref", "bs_sub"):
        f = getattr(L, "dada2hip_result_" + name)
        f.argtypes = [vp]
        f.restype = C.POINTER(C.c_char)
    L.dada2hip_result_stats.argtypes = [vp, C.POINTER(CStats)]
    L.dada2hip_result_free.argtypes = [vp]
    L.dada2hip_nwalign.argtypes = [cp, cp, ip, ip, ip, ip, ip, ip, ip, cp, cp, cp, C.c_size_t]
    L.dada2hip_nwvec.argtypes = [ip, C.POINTER(cp), C.POINTER(cp), ip, ip, ip, ip, ip, ip, C.POINTER(cp), cp, C.c_size_t]
    L.dada2hip_sample_compare.argtypes = [vp, ip, vp, ip, C.POINTER(COpts), C.c_double, vp, vp, vp, vp,
                                          C.POINTER(CStats), cp, C.c_size_t]
    L.dada2hip_calc_pA.argtypes = [ip, vp, vp, vp, ip, vp, cp, C.c_size_t]
    L.dada2hip_run_multi.argtypes = [ip, C.POINTER(CSampleInput), vp, ip, C.POINTER(COpts), ip, vp, C.POINTER(vp), cp,
                                     C.c_size_t]
    L.dada2hip_table_bimera2.argtypes = [ip, ip, vp, C.POINTER(cp), C.c_double, ip, ip, ip, ip, ip, ip, ip, ip, vp, vp, cp, C.c_size_t]
    L.dada2hip_is_bimera.argtypes = [cp, ip, C.POINTER(cp), ip, ip, ip, ip, ip, ip, ip, C.POINTER(ip), cp, C.c_size_t]
    L.dada2hip_bimera_pairs.argtypes = [ip, C.POINTER(cp), C.POINTER(cp), ip, ip, ip, ip, ip, ip, C.c_void_p, cp, C.c_size_t]
    L.dada2hip_derep_fastq.argtypes = [cp, C.c_int64, ip, C.POINTER(vp), cp, C.c_size_t]
    for name, rt in (("nuniques", ip), ("nreads", C.c_int64), ("maxlen", ip), ("seqs", C.POINTER(cp)),
                     ("abundances", C.POINTER(C.c_int32)), ("quals", C.POINTER(C.c_double)), ("map", C.POINTER(C.c_int32))):
        f = getattr(L, "dada2hip_derep_" + name)
        f.argtypes = [vp]
        f.restype = rt
    L.dada2hip_derep_free.argtypes = [vp]
    L.dada2hip_derep_free.restype = None
    L.dada2hip_sample_from_derep.argtypes = [vp, vp, ip, C.POINTER(vp), cp, C.c_size_t]
    L.dada2hip_merge_pairs.argtypes = [C.c_int64, vp, vp, ip, C.POINTER(cp), vp, ip, C.POINTER(cp), vp, ip, ip, ip, ip, ip,
                                       C.POINTER(vp), cp, C.c_size_t]
    L.dada2hip_mergers_nrow.argtypes = [vp]
    L.dada2hip_mergers_nrow.restype = ip
    L.dada2hip_mergers_sequence.argtypes = [vp, ip]
    L.dada2hip_mergers_sequence.restype = cp
    for name in ("abundance", "forward", "reverse", "nmatch", "nmismatch", "nindel", "prefer", "accept"):
        f = getattr(L, "dada2hip_mergers_" + name)
        f.argtypes = [vp]
        f.restype = C.POINTER(C.c_int32)
    L.dada2hip_mergers_free.argtypes = [vp]
    L.dada2hip_mergers_free.restype = None
    L.dada2hip_trim_cache.argtypes = []
    L.dada2hip_trim_cache.restype = None
    _lib = L
    return L


def check(rc, errbuf):
    if rc != 0:
        raise Dada2HipError(rc, errbuf.value.decode(errors="replace"))
