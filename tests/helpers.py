"""Shared comparison helpers for the parity tests."""
import numpy as np

from dada2_amd.opts import DadaResult

P_RTOL = 1e-10  # BASELINE.json north_star: p-values within 1e-10, everything else bit-exact


def _same_float(a, b, rtol):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), "NaN/NA pattern differs"
    x, y = a[~na], b[~nb]
    if rtol == 0:
        assert np.array_equal(x, y), np.max(np.abs(x - y)) if x.size else 0
    else:
        tiny = 5e-324 * 4  # denormal quantisation
        ok = np.abs(x - y) <= rtol * np.maximum(np.abs(x), np.abs(y)) + tiny
        assert ok.all(), ("max rel err", float(np.max(np.abs(x - y)[~ok] / np.maximum(np.abs(x), np.abs(y))[~ok])))


def assert_results_equal(a: DadaResult, b: DadaResult, p_rtol=P_RTOL, check_birth_from=True, exact_float=False):
    """Bit-exact on sequences / indices / counts / membership; p-values within p_rtol."""
    rt = 0 if exact_float else p_rtol
    assert a.clustering["sequence"] == b.clustering["sequence"]
    for c in ("abundance", "n0", "n1", "nunq", "birth_ham") + (("birth_from",) if check_birth_from else ()):
        assert np.array_equal(a.clustering[c], b.clustering[c]), c
    for c in ("pval", "birth_pval"):
        _same_float(a.clustering[c], b.clustering[c], rt)
    # birth_fold = reads/(lambda*parent_reads), birth_qave = integer mean: fp64-exact given exact lambda
    _same_float(a.clustering["birth_fold"], b.clustering["birth_fold"], 0)
    _same_float(a.clustering["birth_qave"], b.clustering["birth_qave"], 0)
    assert np.array_equal(a.birth_subs["pos"], b.birth_subs["pos"])
    assert a.birth_subs["ref"] == b.birth_subs["ref"] and a.birth_subs["sub"] == b.birth_subs["sub"]
    assert np.array_equal(a.birth_subs["clust"], b.birth_subs["clust"])
    _same_float(a.birth_subs["qual"], b.birth_subs["qual"], 0)
    assert np.array_equal(a.subqual, b.subqual)
    _same_float(a.clusterquals, b.clusterquals, 0)
    assert np.array_equal(a.map, b.map)
    _same_float(a.pval, b.pval, rt)
