"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
(a) the committed goldens the reference itself produced and (b) the oracle on seeded inputs.
Bar (BASELINE.json north_star): bit-exact sequences / indices / partition membership / counts,
p-values within 1e-10 relative."""
import os

import numpy as np
import pytest

from helpers import (BAND_OPTION_CASES, BASE_OPTION_CASES, GOLDEN, HOMO_OPTION_CASES, SCORE_OPTION_CASES, WHOLE_PATH_CASES, P_RTOL,
                     assert_results_equal, case_inputs, homopolymer_sample, load_input, long_read_sample, seeded_option_sample, tperr1)
from dada2_amd.io import extend_err
from dada2_amd.opts import DadaOpts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dada2_amd import api as a
    return a


@pytest.mark.parametrize("nw_kernel", ["coop", "lane", "wide"])
@pytest.mark.parametrize("name", WHOLE_PATH_CASES)
def test_whole_path_matches_reference_goldens(api, oracle_c, name, nw_kernel, monkeypatch):
    """Both NW kernels (cooperative anti-diagonal k_nw_ad / lane-per-alignment k_nw) must give the
    reference's answer; DADA2HIP_NW_KERNEL forces the choice the driver otherwise makes by batch size."""
    monkeypatch.setenv("DADA2HIP_NW_KERNEL", nw_kernel)
    d, err, pri, opts, exp, meta = case_inputs(name)
    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts)
    assert got.nclust == meta["nclust"]
    assert_results_equal(got, exp, p_rtol=P_RTOL, check_birth_from="priors" not in meta)
    want = oracle_c.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts)
    assert_results_equal(got, want, p_rtol=P_RTOL, check_birth_from="priors" not in meta)


@pytest.mark.parametrize("fq,band", [("sam1F", 16), ("samPB", 32)])
def test_compare_round_matches_reference(api, fq, band):
    """lambda / hamming / class of one b_compare round against the reference's sub_new + compute_lambda_ts."""
    d = load_input(fq)
    err = extend_err(tperr1(), int(np.ceil(np.nanmax(d.quals))))
    rows = np.load(os.path.join(GOLDEN, f"compare_{fq}.npy"))
    o = DadaOpts(BAND_SIZE=band)
    smp = api.Sample.from_derep(d)
    try:
        for c in sorted(set(rows[:, 0].astype(int))):
            lam, ham, cls, st = smp.compare(c, err, o, kdist_cutoff=0.42)
            sel = rows[rows[:, 0] == c]
            r = sel[:, 1].astype(int)
            assert np.array_equal(lam[r], sel[:, 2]), "lambda bits differ"
            want_ham = np.where(sel[:, 3] < 0, 0xFFFFFFFF, sel[:, 3]).astype(np.uint32)
            assert np.array_equal(ham[r], want_ham)
            shrouded = sel[:, 3] < 0
            assert np.array_equal(cls[r] == 1, shrouded)
            gapless = (~shrouded) & (sel[:, 4] == sel[:, 5])
            assert np.array_equal(cls[r] == 2, gapless)
    finally:
        smp.close()


def test_nwvec_matches_reference_alignments(api):
    z = np.load(os.path.join(GOLDEN, "nwalign_pairs.npz"))
    s1, s2, band = [str(x) for x in z["s1"]], [str(x) for x in z["s2"]], z["band"]
    for b in sorted(set(band.tolist())):
        idx = np.nonzero(band == b)[0]
        got = api.nwvec([s1[i] for i in idx], [s2[i] for i in idx], 5, -4, -8, int(b))
        for k, i in enumerate(idx):
            assert got[k] == (str(z["al0"][i]), str(z["al1"][i])), (i, b)
    assert api.nwalign(s1[0], s2[0], band=16) == (str(z["al0"][0]), str(z["al1"][0]))


def _sample(seed, n, L=120, G=8, Lmin=None, indel=0.0):
    from dada2_amd.synth import make_sample
    return make_sample(tperr1(), n, L=L, G=G, seed=seed, Lmin=Lmin, indel_rate=indel, chunk=4000)


_ids = lambda v: str(v) if isinstance(v, int) else ("-".join(f"{k}={x}" for k, x in v.items()) or "default")


@pytest.mark.parametrize("seed,kw", BASE_OPTION_CASES + BAND_OPTION_CASES, ids=_ids)
def test_seeded_samples_match_oracle(api, oracle_c, seed, kw, monkeypatch):
    monkeypatch.setenv("DADA2HIP_NW_KERNEL", ("lane", "coop", "wide")[seed % 3])
    d, pri = seeded_option_sample(seed)
    o = DadaOpts(**kw)
    got = api.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o)
    want = oracle_c.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o)
    assert_results_equal(got, want, p_rtol=P_RTOL, check_birth_from=pri is None)


@pytest.mark.parametrize("nw_kernel", ["coop", "lane", "wide"])
@pytest.mark.parametrize("seed,kw", SCORE_OPTION_CASES, ids=_ids)
def test_user_alignment_scores_and_sse1_on_every_aligner_family(api, oracle_c, oracle_ref, seed, kw, nw_kernel, monkeypatch):
    """MATCH / MISMATCH / GAP_PENALTY are dada() options (/root/reference/R/dada.R:11-13, Rmain.cpp:33-35): they select the
    non-default instances of k_nw_ad (general scores instead of the 5 / -4 / -8 cost domain), here also on the edge geometry,
    through the non-vectorised aligner, with ties-prone small scores, and next to SSE = 1 (the 16-bit k-mer screen,
    nwalign_endsfree.cpp:27-28).  Each case on all three aligner families, against the restatement AND the reference itself."""
    monkeypatch.setenv("DADA2HIP_NW_KERNEL", nw_kernel)
    d, pri = seeded_option_sample(seed)
    o = DadaOpts(**kw)
    got = api.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o)
    want = oracle_c.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o)
    assert_results_equal(got, want, p_rtol=P_RTOL)
    assert_results_equal(got, oracle_ref.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o), p_rtol=P_RTOL)


@pytest.mark.parametrize("route", ["anti-diagonal", "lane"])
@pytest.mark.parametrize("seed,kw", HOMO_OPTION_CASES, ids=_ids)
def test_homopolymer_gap_penalty_on_homopolymer_rich_reads(api, oracle_c, seed, kw, route, monkeypatch):
    """HOMOPOLYMER_GAP_PENALTY (R/dada.R:14, nwalign_endsfree_homo nwalign_endsfree.cpp:220-396; the 454 / Ion Torrent setting) on
    reads whose errors are run-length changes of homopolymers - a sample on which the option changes the result
    (test_oracle.py pins the oracle to the reference on the same cases).  Since round 4 these runs take k_nw_ad<.., HOMO>, engine
    v2 and the persistent tail; DADA2HIP_AD_HOMO=0 = the lane kernels with the classic engine."""
    if route == "lane":
        monkeypatch.setenv("DADA2HIP_AD_HOMO", "0")
    d = homopolymer_sample(seed)
    err = extend_err(tperr1(), 40)
    o = DadaOpts(**kw)
    got = api.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    assert_results_equal(got, oracle_c.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o), p_rtol=P_RTOL)
    assert (got.stats["tail_launches"] > 0) == (route == "anti-diagonal")


@pytest.mark.parametrize("nw_kernel", ["auto", "lane", "wide"])
def test_reads_longer_than_2047_nt(api, oracle_ref, nw_kernel, monkeypatch):
    """2.1-2.3 kb reads, band 32, PacBio qualities: longer than one block of k_nw_ad stages (nw_ad_lds_bytes returns 0), so
    the rounds take k_nw_adw / the lane kernels.  Against the reference itself."""
    if nw_kernel != "auto":
        monkeypatch.setenv("DADA2HIP_NW_KERNEL", nw_kernel)
    d, err = long_read_sample()
    assert max(map(len, d.seqs)) > 2047
    o = DadaOpts(BAND_SIZE=32)
    got = api.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    want = oracle_ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    assert got.nclust == want.nclust > 8
    assert_results_equal(got, want, p_rtol=P_RTOL)


def test_full_operon_length_reads_4500_nt(api, oracle_ref):
    """4.3-4.5 kb reads (PacBio rRNA-operon amplicons; the reference accepts up to 9 998 nt, dada.h:24)."""
    d, err = long_read_sample(seed=43, n=200, L=4500, Lmin=4300, chunk=2500)
    o = DadaOpts(BAND_SIZE=32)
    got = api.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    oracle_ref.set_threads(os.cpu_count() or 1)
    try:
        want = oracle_ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o, multithread=True)
    finally:
        oracle_ref.set_threads(1)
    assert got.nclust == want.nclust > 8
    assert_results_equal(got, want, p_rtol=P_RTOL)


@pytest.mark.parametrize("nw_kernel", ["auto", "lane"])
@pytest.mark.parametrize("L,Lmin,band", [(300, 200, 32), (300, 160, 32), (260, 100, 64)])
def test_ragged_wide_band_classes(api, oracle_c, L, Lmin, band, nw_kernel, monkeypatch):
    """Ragged long-read shaped inputs: band + length spread of 165 / 205 / 289 cells -> the wide anti-diagonal kernel
    with 21 / 32 / 64 lanes per alignment (auto), and the 193- / 257-cell register classes of the lane kernel and the
    generic any-width kernel (lane)."""
    if nw_kernel != "auto":
        monkeypatch.setenv("DADA2HIP_NW_KERNEL", nw_kernel)
    d = _sample(40 + band + Lmin, 400, L=L, G=8, Lmin=Lmin, indel=1e-3)
    o = DadaOpts(BAND_SIZE=band)
    got = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, o)
    want = oracle_c.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, o)
    assert_results_equal(got, want, p_rtol=P_RTOL)


def _golden_cases_in_subprocess(env_extra, names):
    import subprocess, sys
    code = (
        "import numpy as np, sys\n"
        "root = %r\n"
        "sys.path[:0] = [root, root + '/tests']\n"
        "from helpers import case_inputs, assert_results_equal, P_RTOL\n"
        "from dada2_amd import api\n"
        "for name in %r:\n"
        "    d, err, pri, o, exp, meta = case_inputs(name)\n"
        "    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)\n"
        "    assert_results_equal(got, exp, p_rtol=P_RTOL, check_birth_from=pri is None)\n"
        "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), tuple(names))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_host_applied_births_equal_device_applied():
    """Unambiguous births are normally applied on the device (k_auto_birth) with the next round launched behind them;
    DADA2HIP_NO_AUTOBIRTH=1 keeps every decision on the host (the path ties and prior births always take)."""
    _golden_cases_in_subprocess({"DADA2HIP_NO_AUTOBIRTH": "1"}, ("sam1F_default", "sam1F_priors", "synth3000_default"))


def test_comparison_store_growth():
    """The device-resident comparison store (Bi::comp of every partition) starts at 4 N entries and doubles on demand;
    with the first allocation forced down to N + 16 it has to grow (and be copied) several times in a run."""
    _golden_cases_in_subprocess({"DADA2HIP_NODE_CAP": "1"}, ("sam1F_default", "sam2F_nogreedy", "synth3000_default"))


def test_plain_shuffle_loop_equals_speculative_round_tail(oracle_c):
    """The round tail normally enqueues shuffle + a check-only second shuffle + p-update + bud speculatively; the plain
    loop of Rmain.cpp:320-325 (one shuffle per device round trip, snapshot refreshed by a copy) is what it falls back to
    when the second shuffle still moves uniques.  Both must give the reference's result (run in a subprocess: the knob is
    read once per process)."""
    import subprocess, sys
    code = (
        "import numpy as np, sys\n"
        "root = %r\n"
        "sys.path[:0] = [root, root + '/tests']\n"
        "from helpers import case_inputs, assert_results_equal, P_RTOL\n"
        "from dada2_amd import api\n"
        "for name in ('sam1F_default', 'sam2F_nogreedy', 'synth3000_default'):\n"
        "    d, err, pri, o, exp, meta = case_inputs(name)\n"
        "    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)\n"
        "    assert_results_equal(got, exp, p_rtol=P_RTOL, check_birth_from=pri is None)\n"
        "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DADA2HIP_NO_SPECULATION="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_resident_sample_reuse_and_selfconsist(api, oracle_c):
    """selfConsist loop (R/dada.R:256-405): the resident sample is reused across passes with only err
    changing; every pass must equal the oracle run with the same err."""
    d = _sample(31, 1500, L=150, G=16)
    res, err_out, errs = api.dada(d, None, self_consist=True, opts=DadaOpts(OMEGA_C=0, MAX_CONSIST=4))
    assert len(errs) >= 2
    want = oracle_c.dada_uniques(d.seqs, d.abundances, None, errs[-1], d.quals, DadaOpts(OMEGA_C=0))
    assert_results_equal(res, want, p_rtol=P_RTOL)


def test_device_pvalue_kernel(api, oracle_c):
    rows = np.load(os.path.join(GOLDEN, "ppois_grid.npy"))
    reads = rows[:, 0].astype(np.int32) + 1
    for prior, col in ((0, 4), (1, 5)):
        got = api.calc_pA_device(reads, rows[:, 1], np.full(reads.size, prior, dtype=np.uint8))
        want = rows[:, col]
        ok = np.isfinite(want) & (want > 1e-300)
        rel = np.abs(got[ok] - want[ok]) / want[ok]
        assert rel.max() < P_RTOL, rel.max()
        assert (got[want == 0] == 0).all()


def test_size_independent_properties_at_scale(api):
    """Config-2-like size (bench workload scaled down): properties that hold for any correct run."""
    d = _sample(77, 20000, L=250, G=64)
    r = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts())
    C = r.nclust
    m = r.map
    assert ((m >= 1) & (m <= C) | (m == -2 ** 31)).all()
    # abundance of each partition = reads of the corrected uniques mapped to it; nunq likewise
    ok = m > 0
    assert np.array_equal(np.bincount(m[ok] - 1, weights=d.abundances[ok], minlength=C).astype(np.int64), r.clustering["abundance"].astype(np.int64))
    assert np.array_equal(np.bincount(m[ok] - 1, minlength=C), r.clustering["nunq"])
    # every centre maps to its own partition, has p = 1, and its sequence is the partition's sequence
    cen = r.stats["center"]
    assert np.array_equal(m[cen], np.arange(1, C + 1)) and (r.pval[cen] == 1.0).all()
    assert [d.seqs[c] for c in cen] == r.clustering["sequence"]
    # the transition matrix counts every aligned base of every corrected read once: column sums by row
    # group equal reads x aligned positions; total = sum over corrected uniques of reads * aligned length
    assert r.subqual.sum() > 0 and (r.subqual >= 0).all()
    # births are in decreasing significance only per parent; all birth p-values below OMEGA_A
    assert (r.clustering["birth_pval"][1:] < 1e-40).all()
    # idempotence: a second run on the same resident inputs gives identical output
    r2 = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts())
    assert_results_equal(r, r2, exact_float=True)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 65])
def test_tiny_inputs(api, oracle_c, n):
    """Degenerate sizes: a single unique, fewer uniques than a wave, one over a wave."""
    d = _sample(200 + n, max(n, 8), L=60, G=8)
    seqs, ab, q = d.seqs[:n], d.abundances[:n], d.quals[:n]
    got = api.dada_uniques(seqs, ab, None, tperr1(), q, DadaOpts())
    want = oracle_c.dada_uniques(seqs, ab, None, tperr1(), q, DadaOpts())
    assert_results_equal(got, want, p_rtol=P_RTOL)


def test_unsorted_input_keeps_reference_slot_semantics(api, oracle_c):
    """b_bud skips slot 0 as 'the centre' (cluster.cpp:285) — only true for abundance-sorted input.
    Feed the uniques in reverse order: the reference's quirk must be reproduced, not fixed."""
    d = _sample(301, 500, L=100, G=8)
    seqs, ab, q = d.seqs[::-1], d.abundances[::-1].copy(), d.quals[::-1].copy()
    got = api.dada_uniques(seqs, ab, None, tperr1(), q, DadaOpts())
    want = oracle_c.dada_uniques(seqs, ab, None, tperr1(), q, DadaOpts())
    assert_results_equal(got, want, p_rtol=P_RTOL)


def test_low_complexity_heavy_kmers(api, oracle_c):
    """Sequences whose 5-mers repeat > 63 times exercise the saturated-rank correction of the screen
    (the reference's u8 tables overflow there and fall back to u16, nwalign_endsfree.cpp:23-26)."""
    rng = np.random.default_rng(5)
    base = "A" * 120 + "ACGTTGCA" * 10 + "C" * 100
    seqs, ab = [base], [500]
    for k in range(60):
        s = list(base)
        for _ in range(int(rng.integers(1, 6))):
            p = int(rng.integers(0, len(s)))
            s[p] = "ACGT"[int(rng.integers(0, 4))]
        s = "".join(s)
        if s not in seqs:
            seqs.append(s)
            ab.append(int(rng.integers(1, 40)))
    order = np.argsort(-np.array(ab), kind="stable")
    seqs = [seqs[i] for i in order]
    ab = np.array(ab, dtype=np.int32)[order]
    q = np.full((len(seqs), len(base)), 30.0)
    got = api.dada_uniques(seqs, ab, None, tperr1(), q, DadaOpts())
    want = oracle_c.dada_uniques(seqs, ab, None, tperr1(), q, DadaOpts())
    assert_results_equal(got, want, p_rtol=P_RTOL)


def test_errors_keep_reference_messages(api):
    from dada2_amd import _lib
    d = _sample(302, 100, L=60, G=4)
    with pytest.raises(_lib.Dada2HipError, match="exceeded range of err lookup table"):
        api.dada_uniques(d.seqs, d.abundances, None, tperr1()[:, :20], d.quals, DadaOpts())
    with pytest.raises(_lib.Dada2HipError, match="A/C/G/T"):
        api.dada_uniques(["ACGTNACGTA", "ACGTAACGTA"], [5, 1], None, tperr1(), np.full((2, 10), 30.0), DadaOpts())


def test_nwalign_variants_match_reference(api):
    """C_nwalign's three aligners on the device (src/evaluate.cpp:18-62): nwalign_endsfree, nwalign_endsfree_homo (a gap
    opposite a homopolymer base has its own penalty) and the global nwalign of endsfree=FALSE, on 300 homopolymer-rich
    pairs whose alignments the reference itself produced (tests/golden/make_homo_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "nwalign_variants.npz"))
    for i in range(len(z["s1"])):
        got = api.nwalign(str(z["s1"][i]), str(z["s2"][i]), 5, -4, -8, homo_gap=int(z["homo_gap"][i]), band=int(z["band"][i]),
                          endsfree=bool(z["endsfree"][i]))
        assert got == (str(z["al0"][i]), str(z["al1"][i])), (i, int(z["band"][i]), int(z["homo_gap"][i]), int(z["endsfree"][i]))


def test_work_counters_match_reference_counts(api, oracle_c):
    """The run's comparison counters must equal the reference's own (dada.h:113-114 nalign / nshroud):
    guards against correct-but-wasteful dispatch (e.g. shrouded uniques sent to the aligner)."""
    d, err, pri, opts, exp, meta = case_inputs("sam1F_default")
    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts)
    want = oracle_c.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts)   # serial counters: real alignments only
    st = got.stats
    assert st["ncompare"] - st["nskipped"] == want.stats["nalign"] == 8655
    assert st["nshroud"] == want.stats["nshroud"] == 3032
    # every non-skipped, non-shrouded comparison is exactly one gapless pairing or one NW (+ final pass + births)
    rounds_work = st["ncompare"] - st["nskipped"] - st["nshroud"]
    assert st["nnw"] + st["ngapless"] == rounds_work + d.nraw + (got.nclust - 1)


def test_full_config2_parity_vs_reference_itself(api):
    """BASELINE.json configs[1] at full size (100 k uniques x 250 nt, tperr1): the GPU result against the
    reference's own C++ (oracle/_ref, built in the authoring container and shipped as a binary) on all host
    cores.  Skipped where the prebuilt reference is absent."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not present")
    from dada2_amd.synth import make_sample
    d = make_sample(tperr1(), 100_000, L=250, G=256, seed=20260925 + 2)
    got = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts())
    ref.set_threads(os.cpu_count() or 1)
    want = ref.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts(), multithread=True)
    ref.set_threads(1)
    assert got.nclust == want.nclust > 50
    assert_results_equal(got, want, p_rtol=P_RTOL)


def test_nwvec_on_letters_outside_acgt_matches_the_reference(api, oracle_ref):
    """C_nwvec on N / IUPAC / arbitrary letters (it compares raw bytes, nwalign_vectorized.cpp:165) on the device: each pair's
    letters renumbered into two 2-bit planes, every pair against the reference's own call; > 16 letters and C_nwalign with N
    stay DADA2HIP_ERR_UNSUPPORTED (the latter has no defined behaviour in the reference, evaluate.cpp:28-33)."""
    from helpers import nwvec_letter_cases
    from dada2_amd import _lib
    s1, s2 = nwvec_letter_cases()
    for band, ef, sc in ((16, True, (5, -4, -8)), (-1, True, (5, -4, -8)), (8, False, (5, -4, -8)), (16, True, (1, -1, -2))):
        got = api.nwvec(s1, s2, sc[0], sc[1], sc[2], band, ef)
        for i, (a, b) in enumerate(zip(s1, s2)):
            assert tuple(got[i]) == oracle_ref.nwvec_raw(a, b, sc[0], sc[1], sc[2], band, ef), (band, ef, sc, a, b)
    with pytest.raises(_lib.Dada2HipError) as ei:
        api.nwvec(["ABCDEFGHIJKLMNOPQRS"], ["ABCDEFGHIJKLMNOPQ"])
    assert ei.value.code == 4
    with pytest.raises(_lib.Dada2HipError) as ei:
        api.nwalign("ACGTN", "ACGT")
    assert ei.value.code == 4
