"""The drop-in boundary as CODE: the Rcpp translation unit a maintainer adds to the reference package (INTEGRATION.md,
tests/glue/dada2hip_glue.cpp) compiled against the from-scratch Rcpp model of oracle/shim/Rcpp.h and linked with
libdada2hip.so.  `dada_uniques`, `C_nwalign`, `C_nwvec`, `C_table_bimera2` are called with the reference's exact C++
signatures (/root/reference/src/RcppExports.cpp:18-53, :94, :227, Rmain.cpp:30-47) and the Rcpp::List / CharacterVector /
DataFrame that come back are compared with what the reference's own functions return (oracle/_ref) and with the goldens."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, assert_results_equal, case_inputs, P_RTOL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "tests", "glue")


# the mangled names of the four exports as the reference's own objects define them (nm -D oracle/_ref/libdada2ref.so)
REFERENCE_EXPORTS = (
    "_Z12dada_uniquesSt6vectorINSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEESaIS5_EES_IiSaIiEES_IbSaIbEEN4Rcpp3MatIdEESE_iiibdidddbidiibbbibbibb",
    "_Z9C_nwalignNSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEES4_iiiiib",
    "_Z7C_nwvecSt6vectorINSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEESaIS5_EES7_sssib",
    "_Z15C_table_bimera2N4Rcpp3MatIiEESt6vectorINSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEESaIS8_EEdibiiiii",
)


def build_glue():
    subprocess.check_call(["make", "-s", "-C", GLUE])
    return os.path.join(GLUE, "libdada2glue.so")


def test_integration_md_shows_the_compiled_glue():
    """The stub in INTEGRATION.md IS the translation unit the tests compile: no drift between the document and the code."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)[0]
    assert block == open(os.path.join(GLUE, "dada2hip_glue.cpp")).read()


def test_glue_compiles_and_exports_the_reference_entry_points():
    """g++ only (no GPU): the TU compiles against the Rcpp model and links with libdada2hip.so; the library exports the C++
    symbols with the reference's mangled names, i.e. the signatures match Rmain.cpp:30 / evaluate.cpp:18 /
    nwalign_vectorized.cpp:321 / chimera.cpp:192 character for character."""
    lib = build_glue()
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    mine = {line.split()[-1] for line in syms.splitlines() if line.split()[-1].startswith(("_Z12dada_uniques", "_Z9C_nwalign", "_Z7C_nwvec", "_Z15C_table_bimera2"))}
    assert mine == set(REFERENCE_EXPORTS), mine ^ set(REFERENCE_EXPORTS)
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdada2ref.so")):   # ... which are the names the reference's own objects define
        ref_syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "oracle", "_ref", "libdada2ref.so")],
                                  capture_output=True, text=True, check=True).stdout
        for name in REFERENCE_EXPORTS:
            assert name in ref_syms, name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sam1F_default", "sam1F_priors", "sam2F_nogreedy", "samPB_band32"])
def test_dada_uniques_through_the_rcpp_glue_equals_the_reference(oracle_ref, name):
    build_glue()
    d, err, pri, opts, exp, meta = case_inputs(name)
    got = oracle_ref.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts, flavour="glue")
    want = oracle_ref.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts)
    assert_results_equal(got, want, p_rtol=P_RTOL, check_birth_from=pri is None)
    assert_results_equal(got, exp, p_rtol=P_RTOL, check_birth_from=pri is None)


@pytest.mark.gpu
def test_error_messages_come_back_as_rcpp_stop(oracle_ref):
    build_glue()
    d, err, pri, opts, exp, meta = case_inputs("sam1F_default")
    with pytest.raises(RuntimeError, match="exceeded range of err lookup table"):
        oracle_ref.dada_uniques(d.seqs, d.abundances, None, err[:, :20], d.quals, opts, flavour="glue")


@pytest.mark.gpu
def test_nwalign_and_nwvec_through_the_glue_match_the_reference_goldens():
    L = C.CDLL(build_glue())
    z = np.load(os.path.join(GOLDEN, "nwalign_variants.npz"))
    eb = C.create_string_buffer(512)
    for i in range(0, len(z["s1"]), 3):
        s1, s2 = str(z["s1"][i]).encode(), str(z["s2"][i]).encode()
        o0, o1 = C.create_string_buffer(len(s1) + len(s2) + 2), C.create_string_buffer(len(s1) + len(s2) + 2)
        assert L.glue_nwalign(s1, s2, 5, -4, -8, int(z["homo_gap"][i]), int(z["band"][i]), int(bool(z["endsfree"][i])), o0, o1, eb, 512) == 0, eb.value
        assert (o0.value.decode(), o1.value.decode()) == (str(z["al0"][i]), str(z["al1"][i])), i
    z = np.load(os.path.join(GOLDEN, "nwalign_pairs.npz"))
    idx = np.nonzero(z["band"] == 16)[0][:64]
    n = len(idx)
    a = (C.c_char_p * n)(*[str(z["s1"][i]).encode() for i in idx])
    b = (C.c_char_p * n)(*[str(z["s2"][i]).encode() for i in idx])
    bufs = [C.create_string_buffer(1024) for _ in range(2 * n)]
    out = (C.c_char_p * (2 * n))(*[C.cast(x, C.c_char_p) for x in bufs])
    assert L.glue_nwvec(n, a, b, 5, -4, -8, 16, 1, out, eb, 512) == 0, eb.value
    for k, i in enumerate(idx):
        assert (bufs[2 * k].value.decode(), bufs[2 * k + 1].value.decode()) == (str(z["al0"][i]), str(z["al1"][i])), i


@pytest.mark.gpu
def test_table_bimera2_through_the_glue_matches_the_reference_golden():
    L = C.CDLL(build_glue())
    z = np.load(os.path.join(GOLDEN, "bimera_table.npz"))
    mat = np.asfortranarray(z["mat"].astype(np.int32))
    seqs = [str(s).encode() for s in z["seqs"]]
    nrow, ncol = mat.shape
    sp = (C.c_char_p * ncol)(*seqs)
    nflag, nsam = np.zeros(ncol, dtype=np.int32), np.zeros(ncol, dtype=np.int32)
    eb = C.create_string_buffer(512)
    L.glue_table_bimera2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    for oo in (0, 1):
        assert L.glue_table_bimera2(nrow, ncol, mat.ctypes.data, sp, 1.5, 2, oo, 4, 5, -4, -8, 16, nflag.ctypes.data, nsam.ctypes.data, eb, 512) == 0, eb.value
        assert np.array_equal(nflag, z[f"nflag_oo{oo}_ms16"]) and np.array_equal(nsam, z[f"nsam_oo{oo}_ms16"])
