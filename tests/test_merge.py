"""mergePairs (R/paired.R:92-201): the reference's C helpers compiled in place (oracle/_ref: C_nwalign, C_eval_pair,
C_pair_consensus) pin the plain-C restatement and the committed goldens; the GPU tests run the product
(dada2hip_merge_pairs: unique pairs on the host, every forward x rc(reverse) alignment on the device) against both."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, case_inputs
from merge_cases import OPTION_SETS, make_case
from oracle import cport, merge as omerge

KEYS = ("sequence", "abundance", "forward", "reverse", "nmatch", "nmismatch", "nindel", "prefer", "accept")


def golden_rows(seed, name):
    z = np.load(os.path.join(GOLDEN, "merge_pairs.npz"))
    return json.loads(str(z[f"c{seed}_{name}"]))


def assert_rows_equal(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for k in KEYS:
            assert g[k] == w[k], (k, g, w)


def test_golden_inputs_are_the_committed_cases():
    z = np.load(os.path.join(GOLDEN, "merge_pairs.npz"))
    for seed in (1, 2, 3):
        c = make_case(seed)
        assert list(z[f"c{seed}_seqsF"]) == c["seqsF"] and list(z[f"c{seed}_seqsR"]) == c["seqsR"]
        np.testing.assert_array_equal(z[f"c{seed}_fwd"], c["fwd"])
        np.testing.assert_array_equal(z[f"c{seed}_rev"], c["rev"])


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("name", sorted(OPTION_SETS))
def test_restatement_matches_reference_goldens(seed, name, oracle_c):
    c = make_case(seed)
    rows = omerge.merge_pairs(c["fwd"], c["rev"], c["seqsF"], c["n0F"], c["seqsR"], c["n0R"], cport, return_rejects=True,
                              **OPTION_SETS[name])
    want = golden_rows(seed, name)
    assert_rows_equal(rows, want)
    if name == "default":
        assert any(r["accept"] for r in want) and any(not r["accept"] for r in want)
    if name == "mismatch1":    # (with the default -64 / -64 scores an alignment would rather slide apart than hold a mismatch)
        assert any(r["nmismatch"] + r["nindel"] > 0 for r in want)


def test_helpers_match_the_reference_on_random_alignments(oracle_c, oracle_ref):
    rng = np.random.default_rng(7)
    for _ in range(300):
        a = "".join(rng.choice(list("ACGT"), size=int(rng.integers(20, 80))))
        b = list(a[int(rng.integers(0, 15)):])
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(len(b)))
            b[p] = "ACGT"[int(rng.integers(4))]
        if rng.random() < 0.4 and len(b) > 10:
            del b[int(rng.integers(len(b)))]
        b = "".join(b) + "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 12))))
        sc = (1, -64, -64) if rng.random() < 0.5 else (1, -8, -8)
        a1, a2 = oracle_ref.C_nwalign(a, b, *sc, None, -1, True)
        assert (a1, a2) == cport.nwalign(a, b, *sc, band=-1)
        assert oracle_ref.eval_pair(a1, a2) == cport.eval_pair(a1, a2)
        for prefer in (1, 2):
            for trim in (False, True):
                assert oracle_ref.pair_consensus(a1, a2, prefer, trim) == cport.pair_consensus(a1, a2, prefer, trim)


def _call_merge(c, **kw):
    import ctypes as C
    from dada2_amd import _lib
    L = _lib.lib()
    NA = np.iinfo(np.int32).min
    fwd = np.where(c["fwd"] > 0, c["fwd"], NA).astype(np.int32)
    rev = np.where(c["rev"] > 0, c["rev"], NA).astype(np.int32)
    aF = (C.c_char_p * len(c["seqsF"]))(*[s.encode() for s in c["seqsF"]])
    aR = (C.c_char_p * len(c["seqsR"]))(*[s.encode() for s in c["seqsR"]])
    eb = C.create_string_buffer(512)
    h = C.c_void_p()
    rc = L.dada2hip_merge_pairs(len(fwd), fwd.ctypes.data, rev.ctypes.data, len(c["seqsF"]), aF, c["n0F"].ctypes.data, len(c["seqsR"]), aR,
                                c["n0R"].ctypes.data, kw.get("min_overlap", 12), kw.get("max_mismatch", 0), int(kw.get("trim_overhang", False)),
                                int(kw.get("just_concatenate", False)), 0, C.byref(h), eb, 512)
    if rc:
        return rc, eb.value.decode(), None
    n = L.dada2hip_mergers_nrow(h)
    col = {k: np.ctypeslib.as_array(getattr(L, "dada2hip_mergers_" + k)(h), (n,)).copy() if n else np.zeros(0, np.int32)
           for k in ("abundance", "forward", "reverse", "nmatch", "nmismatch", "nindel", "prefer", "accept")}
    rows = [{"sequence": L.dada2hip_mergers_sequence(h, i).decode(), **{k: int(col[k][i]) for k in col}} for i in range(n)]
    L.dada2hip_mergers_free(h)
    for r in rows:
        r["prefer"] = None if r["prefer"] == NA else r["prefer"]
        r["accept"] = bool(r["accept"])
    return 0, "", rows


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_host_side_of_merge_pairs_without_a_device(seed):
    """What dada2hip_merge_pairs does on the host alone: the unique-pair bookkeeping and justConcatenate (no alignment, so no
    device is needed), the reference's input check, and the empty case."""
    c = make_case(seed)
    rc, msg, rows = _call_merge(c, just_concatenate=True)
    assert rc == 0, msg
    assert_rows_equal(rows, golden_rows(seed, "concat"))
    bad = dict(c, fwd=np.where(c["fwd"] > 0, c["fwd"] + 100, c["fwd"]).astype(np.int32))
    rc, msg, _ = _call_merge(bad, just_concatenate=True)
    assert rc == 1 and "Non-corresponding derep-class and dada-class objects." in msg
    none = dict(c, fwd=np.full_like(c["fwd"], -1))
    rc, msg, rows = _call_merge(none)
    assert rc == 0 and rows == []


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("name", sorted(OPTION_SETS))
def test_device_merge_matches_goldens_and_restatement(seed, name):
    import ctypes as C
    from dada2_amd import _lib
    c = make_case(seed)
    L = _lib.lib()
    NA = np.iinfo(np.int32).min
    fwd = np.where(c["fwd"] > 0, c["fwd"], NA).astype(np.int32)
    rev = np.where(c["rev"] > 0, c["rev"], NA).astype(np.int32)
    kw = dict(min_overlap=12, max_mismatch=0, trim_overhang=False, just_concatenate=False)
    kw.update(OPTION_SETS[name])
    aF = (C.c_char_p * len(c["seqsF"]))(*[s.encode() for s in c["seqsF"]])
    aR = (C.c_char_p * len(c["seqsR"]))(*[s.encode() for s in c["seqsR"]])
    eb = C.create_string_buffer(512)
    h = C.c_void_p()
    rc = L.dada2hip_merge_pairs(len(fwd), fwd.ctypes.data, rev.ctypes.data, len(c["seqsF"]), aF, c["n0F"].ctypes.data, len(c["seqsR"]), aR,
                                c["n0R"].ctypes.data, kw["min_overlap"], kw["max_mismatch"], int(kw["trim_overhang"]),
                                int(kw["just_concatenate"]), 0, C.byref(h), eb, 512)
    assert rc == 0, eb.value
    n = L.dada2hip_mergers_nrow(h)
    col = {k: np.ctypeslib.as_array(getattr(L, "dada2hip_mergers_" + k)(h), (n,)).copy()
           for k in ("abundance", "forward", "reverse", "nmatch", "nmismatch", "nindel", "prefer", "accept")}
    rows = [{"sequence": L.dada2hip_mergers_sequence(h, i).decode(), **{k: int(col[k][i]) for k in col}} for i in range(n)]
    L.dada2hip_mergers_free(h)
    for r in rows:
        r["prefer"] = None if r["prefer"] == NA else r["prefer"]
        r["accept"] = bool(r["accept"])
    assert_rows_equal(rows, golden_rows(seed, name))
    assert_rows_equal(rows, omerge.merge_pairs(c["fwd"], c["rev"], c["seqsF"], c["n0F"], c["seqsR"], c["n0R"], cport,
                                               return_rejects=True, **OPTION_SETS[name]))


@pytest.mark.gpu
def test_whole_path_sam1_forward_reverse_merge():
    """The reference's own example (R/paired.R:86-89): dada() on sam1F and sam1R, then mergePairs on the two results."""
    from dada2_amd import api
    from dada2_amd.io import Derep
    maps = np.load(os.path.join(GOLDEN, "sam1_maps.npz"))
    res, dereps = {}, {}
    for fq, case in (("sam1F", "sam1F_default"), ("sam1R", "sam1R_default")):
        d, err, pri, opts, exp, meta = case_inputs(case)
        res[fq] = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts, device=0)
        dereps[fq] = Derep(d.seqs, d.abundances, d.quals, maps[fq])
    got = api.merge_pairs(res["sam1F"], dereps["sam1F"], res["sam1R"], dereps["sam1R"], return_rejects=True)
    fwd = np.asarray(res["sam1F"].map)[maps["sam1F"]]
    rev = np.asarray(res["sam1R"].map)[maps["sam1R"]]
    want = omerge.merge_pairs(fwd, rev, list(res["sam1F"].clustering["sequence"]), res["sam1F"].clustering["n0"],
                              list(res["sam1R"].clustering["sequence"]), res["sam1R"].clustering["n0"], cport, return_rejects=True)
    assert_rows_equal(got, want)
    assert sum(r["abundance"] for r in got if r["accept"]) > 1000     # most of the 1 500 read pairs of the fixture merge
    ok = [r for r in api.merge_pairs(res["sam1F"], dereps["sam1F"], res["sam1R"], dereps["sam1R"])]
    assert ok == [r for r in got if r["accept"]]
