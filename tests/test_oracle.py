"""CPU tests (-m "not gpu"): pin the oracle.

1. oracle/dada_oracle.c (plain-C restatement) against the committed goldens that the
   REFERENCE ITSELF produced (tests/golden/make_golden.py via oracle/_ref) — bit-exact,
   p-values included, since both sides share oracle/rmath_ppois.c.
2. where oracle/_ref is present, restatement vs reference live on fresh seeded inputs.
3. the ppois restatement against 60-digit truth (golden) and scipy.
"""
import os

import numpy as np
import pytest

from helpers import (BASE_OPTION_CASES, GOLDEN, SCORE_OPTION_CASES, WHOLE_PATH_CASES, assert_results_equal, case_inputs,
                     long_read_sample, seeded_option_sample, tperr1)
from dada2_amd.opts import DadaOpts


@pytest.mark.parametrize("name", WHOLE_PATH_CASES)
def test_restatement_matches_reference_goldens(oracle_c, name):
    d, err, pri, opts, exp, meta = case_inputs(name)
    got = oracle_c.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts)
    assert got.nclust == meta["nclust"]
    assert_results_equal(got, exp, exact_float=True, check_birth_from="priors" not in meta)


def test_known_outcomes_from_survey(oracle_c):
    """SURVEY.md §4 probe: sam1F + tperr1 -> 10 partitions, 8655 comparisons / 3032 shrouded."""
    d, err, pri, opts, exp, meta = case_inputs("sam1F_default")
    got = oracle_c.dada_uniques(d.seqs, d.abundances, None, err, d.quals, opts)
    assert got.clustering["abundance"].tolist() == [543, 376, 160, 61, 47, 69, 9, 13, 82, 68]
    assert (got.stats["nalign"], got.stats["nshroud"]) == (8655, 3032)
    d, err, pri, opts, exp, meta = case_inputs("samPB_band32")
    got = oracle_c.dada_uniques(d.seqs, d.abundances, None, err, d.quals, opts)
    assert got.clustering["abundance"].tolist() == [59, 107, 53, 68, 53, 27, 34, 18, 8, 13, 18, 12]


def test_nwalign_goldens(oracle_c):
    z = np.load(os.path.join(GOLDEN, "nwalign_pairs.npz"))
    for s1, s2, band, a0, a1 in zip(z["s1"], z["s2"], z["band"], z["al0"], z["al1"]):
        assert oracle_c.nwalign(str(s1), str(s2), 5, -4, -8, int(band)) == (str(a0), str(a1))


@pytest.mark.parametrize("fq,band", [("sam1F", 16), ("samPB", 32)])
def test_compare_goldens(oracle_c, fq, band):
    from helpers import load_input
    from dada2_amd.io import extend_err
    d = load_input(fq)
    err = extend_err(tperr1(), int(np.ceil(np.nanmax(d.quals))))
    rows = np.load(os.path.join(GOLDEN, f"compare_{fq}.npy"))
    o = DadaOpts(BAND_SIZE=band)
    for c, r, lam, ham, kd, ko in rows[:: (1 if fq == "sam1F" else 2)]:
        c, r = int(c), int(r)
        got = oracle_c.compare(d.seqs[c], d.quals[c, :len(d.seqs[c])], d.seqs[r], d.quals[r, :len(d.seqs[r])], err, o,
                               kdist_cutoff=0.42)
        assert got == (lam, int(ham), kd, ko)


def test_input_validation(oracle_c):
    err = tperr1()
    with pytest.raises(RuntimeError, match="Zero input"):
        oracle_c.dada_uniques([], [], None, err, None)
    with pytest.raises(RuntimeError, match="kmer-size"):
        oracle_c.dada_uniques(["ACGT"], [1], None, err, None)


def test_ppois_against_truth_and_scipy(oracle_c):
    from scipy.special import pdtrc
    rows = np.load(os.path.join(GOLDEN, "ppois_grid.npy"))
    x, lam, truth, ref_p = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
    got = np.array([oracle_c.ppois_upper(a, b) for a, b in zip(x, lam)])
    assert np.array_equal(got, ref_p)                      # same bits as when the goldens were made
    normal = truth > 2.3e-308
    rel = np.abs(got[normal] - truth[normal]) / truth[normal]
    # R's algorithm redoes results < 1e-292 in log space (pgamma.c), losing ~|log p|*eps there
    main = truth[normal] > 1e-290
    # dpois_raw = exp(-stirlerr - bd0)/sqrt(2 pi x): conditioning ~ |log p| * eps, so deep tails carry ~1e-12
    assert rel[main].max() < 2e-12, rel[main].max()
    shallow = truth[normal] > 1e-20
    assert rel[shallow].max() < 1e-13, rel[shallow].max()
    assert rel[~main].max() < 2e-11, rel[~main].max()
    assert (got[truth == 0.0] == 0.0).all()
    sp = pdtrc(x[normal][main], lam[normal][main])
    ok = sp > 1e-300
    assert (np.abs(sp[ok] - got[normal][main][ok]) / got[normal][main][ok]).max() < 1e-11
    # calc_pA columns (conditional / unconditional), reference pval.cpp:44-64
    for (a, b, pc, pu) in rows[::7, [0, 1, 4, 5]]:
        assert oracle_c.calc_pA(int(a) + 1, b, False) == pc or (np.isnan(pc) and np.isnan(oracle_c.calc_pA(int(a) + 1, b, False)))
        assert oracle_c.calc_pA(int(a) + 1, b, True) == pu


# ---------------------------------------------------------------------------------------------
# live cross-checks against the reference (only where oracle/_ref has been built)
def _random_sample(seed, n=600, L=120, G=8, Lmin=None, indel=0.0):
    from dada2_amd.synth import make_sample
    return make_sample(tperr1(), n, L=L, G=G, seed=seed, Lmin=Lmin, indel_rate=indel, chunk=2000)


@pytest.mark.parametrize("seed,kw", BASE_OPTION_CASES + SCORE_OPTION_CASES, ids=lambda v: str(v) if isinstance(v, int) else "-".join(f"{k}={x}" for k, x in v.items()) or "default")
def test_restatement_vs_reference_live(oracle_c, oracle_ref, seed, kw):
    """The seeded option sweep (tests/helpers.py): restatement == reference, serial and multithreaded - among them the user
    alignment scores and SSE = 1 that select code paths of their own on both sides."""
    d, pri = seeded_option_sample(seed, n=600, chunk=2000)
    o = DadaOpts(**kw)
    for mt in (False, True):
        a = oracle_ref.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o, multithread=mt)
        b = oracle_c.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o, multithread=mt)
        assert_results_equal(a, b, exact_float=True, check_birth_from=pri is None)


def test_restatement_vs_reference_reads_longer_than_2047(oracle_c, oracle_ref):
    """2.1-2.3 kb reads (PacBio-style qualities, band 32): the checker of the -m gpu long-read case, pinned to the reference."""
    d, err = long_read_sample()
    o = DadaOpts(BAND_SIZE=32)
    a = oracle_ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    b = oracle_c.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    assert a.nclust > 8 and max(map(len, d.seqs)) > 2047
    assert_results_equal(a, b, exact_float=True)


# NOTE: there is no "no quality matrix" case to pin: with a 0-row quals matrix the reference
# itself crashes (error.cpp:158 reads raw->qual[pos1] unconditionally), and R always passes
# quals (R/dada.R:337) — the product therefore rejects quals == NULL up front.


def test_nwalign_vs_both_reference_aligners(oracle_c, oracle_ref):
    rng = np.random.default_rng(99)
    for k in range(1500):
        L1 = int(rng.integers(6, 200))
        a = rng.integers(0, 4, L1)
        b = a.copy() if k % 4 else rng.integers(0, 4, int(rng.integers(6, 200)))
        for _ in range(int(rng.integers(0, 6))):
            p = int(rng.integers(0, len(b)))
            r = rng.random()
            b = np.delete(b, p) if (r < 0.3 and len(b) > 8) else (np.insert(b, p, rng.integers(0, 4)) if r < 0.6 else b)
            if r >= 0.6:
                b[p] = rng.integers(0, 4)
        s1 = "".join("ACGT"[i] for i in a)
        s2 = "".join("ACGT"[i] for i in b)
        band = int(rng.choice([1, 2, 3, 8, 15, 16, 17, 32, -1]))
        g = int(rng.choice([-8, -2, -12]))
        want = oracle_ref.nwalign(s1, s2, 5, -4, g, band, "vectorized")
        assert want == oracle_ref.nwalign(s1, s2, 5, -4, g, band, "endsfree")
        assert oracle_c.nwalign(s1, s2, 5, -4, g, band) == want


def test_homopolymer_gap_restatement_matches_the_reference_on_homopolymer_rich_reads(oracle_c, oracle_ref):
    """nwalign_endsfree_homo inside the whole path (HOMOPOLYMER_GAP_PENALTY, the 454 / Ion Torrent setting): reads whose errors are
    run-length changes of homopolymers, where the option changes the result - restatement vs the reference itself."""
    from helpers import HOMO_OPTION_CASES, homopolymer_sample
    from dada2_amd.io import extend_err
    changed = 0
    for seed, kw in HOMO_OPTION_CASES:
        d = homopolymer_sample(seed)
        err = extend_err(tperr1(), 40)
        want = oracle_ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, DadaOpts(**kw))
        got = oracle_c.dada_uniques(d.seqs, d.abundances, None, err, d.quals, DadaOpts(**kw))
        assert_results_equal(got, want)
        plain = dict(kw); plain.pop("HOMOPOLYMER_GAP_PENALTY")
        other = oracle_c.dada_uniques(d.seqs, d.abundances, None, err, d.quals, DadaOpts(**plain))
        changed += not np.array_equal(other.subqual, got.subqual)
    assert changed == len(HOMO_OPTION_CASES)          # (the sample is one on which the homopolymer penalty matters)


def test_bimera_pair_quantities_restatement_matches_the_reference(oracle_c, oracle_ref):
    """get_lr / get_ham_endsfree per alignment (chimera.cpp:211-293): the restatement against the reference's own functions on
    its own alignments - shared halves, shifts, truncations, indels, unrelated pairs; one-off on and off, four bands, two score sets."""
    from helpers import BIMERA_PAIR_OPTIONS, bimera_pair_cases
    for seed, n, L in ((1, 150, 60), (2, 150, 130), (3, 120, 251)):
        qs, ps = bimera_pair_cases(seed, n, L)
        for oo, ms, sc in BIMERA_PAIR_OPTIONS:
            want = oracle_ref.bimera_pairs(qs, ps, oo, *sc, ms)
            got = oracle_c.bimera_pairs(qs, ps, oo, *sc, ms)
            assert np.array_equal(got, want), (seed, oo, ms, sc, np.nonzero((got != want).any(axis=1))[0][:5])
        assert (want[:, 0] + want[:, 1] > 0).any() and (want[:, 4] > 0).any()
    from helpers import bimera_short_pair_cases
    qs, ps = bimera_short_pair_cases(5, 400)
    for oo, ms in ((True, 16), (False, 16), (True, 4), (True, 1), (True, 40)):
        assert np.array_equal(oracle_c.bimera_pairs(qs, ps, oo, max_shift=ms), oracle_ref.bimera_pairs(qs, ps, oo, max_shift=ms)), (oo, ms)


def test_bimera_restatement_matches_reference_goldens(oracle_c):
    """Bimera identification (src/chimera.cpp): the C restatement against the vectors the reference itself produced
    (tests/golden/make_bimera_golden.py) - the checker of the GPU path's dada2hip_table_bimera2 / dada2hip_is_bimera."""
    z = np.load(os.path.join(GOLDEN, "bimera_table.npz"))
    mat, seqs = z["mat"], [str(s) for s in z["seqs"]]
    for oo in (0, 1):
        for ms in (16, 4):
            nflag, nsam = oracle_c.table_bimera2(mat, seqs, allow_one_off=bool(oo), max_shift=ms)
            assert np.array_equal(nflag, z[f"nflag_oo{oo}_ms{ms}"]) and np.array_equal(nsam, z[f"nsam_oo{oo}_ms{ms}"])
        tot = mat.sum(axis=0)
        for j, s in enumerate(seqs):
            pars = [seqs[k] for k in range(len(seqs)) if tot[k] > 2 * tot[j] and tot[k] > 8]
            assert oracle_c.is_bimera(s, pars, allow_one_off=bool(oo)) == bool(z[f"isbim_oo{oo}"][j]), (j, oo)
    assert z["nflag_oo0_ms16"].max() == 2 and z["nflag_oo1_ms16"].sum() > z["nflag_oo0_ms16"].sum()


def test_restated_aligner_variants_match_the_reference_goldens(oracle_c):
    """nw_general() of the C restatement - nwalign_endsfree, nwalign_endsfree_homo, global nwalign - on the 300 pairs whose
    alignments came out of the reference's own C_nwalign (tests/golden/make_homo_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "nwalign_variants.npz"))
    kinds = set()
    for i in range(len(z["s1"])):
        got = oracle_c.C_nwalign(str(z["s1"][i]), str(z["s2"][i]), 5, -4, -8, int(z["homo_gap"][i]), int(z["band"][i]), bool(z["endsfree"][i]))
        assert got == (str(z["al0"][i]), str(z["al1"][i])), i
        kinds.add((bool(z["endsfree"][i]), int(z["homo_gap"][i]) != -8))
    assert kinds == {(True, False), (True, True), (False, False), (False, True)}


def test_restated_homopolymer_path_matches_the_reference_live(oracle_c, oracle_ref):
    """dada_uniques with HOMOPOLYMER_GAP_PENALTY (raw_align -> nwalign_endsfree_homo) : restatement vs the reference, live."""
    from dada2_amd.opts import DadaOpts
    from dada2_amd.synth import make_sample
    d = make_sample(tperr1(), 700, L=110, G=8, seed=77, Lmin=90, indel_rate=3e-3, chunk=4000)
    for kw in (dict(HOMOPOLYMER_GAP_PENALTY=-1), dict(HOMOPOLYMER_GAP_PENALTY=-2, BAND_SIZE=-1), dict(HOMOPOLYMER_GAP_PENALTY=0, BAND_SIZE=8)):
        o = DadaOpts(**kw)
        a = oracle_c.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, o)
        b = oracle_ref.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, o)
        assert_results_equal(a, b, exact_float=True)
