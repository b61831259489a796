"""GPU parity tests (-m gpu), part 2: the sizes and shapes BASELINE.json's configs name, and the edge cases the
round-1 review found untested — 5'-offset long-read alignments (|i - j| far beyond 127 on the wide anti-diagonal
kernel), unsorted inputs over several seeds, near-tied bud candidates, the batch C entry point, and the RCCL path
of the sample-sharded driver.  Everything goes through the C ABI; the checker is the oracle / the reference binary."""
import os

import numpy as np
import pytest

from helpers import P_RTOL, assert_results_equal, tperr1
from dada2_amd.io import Derep, extend_err
from dada2_amd.opts import DadaOpts

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dada2_amd import api as a
    return a


def _mutate(rng, s, nsub=0, dels=(), ins=()):
    s = list(s)
    for _ in range(nsub):
        p = int(rng.integers(0, len(s)))
        s[p] = "ACGT"[("ACGT".index(s[p]) + int(rng.integers(1, 4))) & 3]
    for p in sorted(dels, reverse=True):
        del s[p]
    for p in sorted(ins, reverse=True):
        s.insert(p, "ACGT"[int(rng.integers(0, 4))])
    return "".join(s)


# ---- 5'-truncated reads: raw = suffix of the centre, the path runs |i - j| = offset cells off the main diagonal ---------
@pytest.mark.parametrize("nw_kernel", ["wide", "lane"])
@pytest.mark.parametrize("L,offsets", [(700, (64, 130, 250, 400)), (1500, (127, 128, 200, 440))])
def test_five_prime_offsets_alignment_level(api, oracle_c, L, offsets, nw_kernel, monkeypatch):
    """One b_compare round (sub_new + compute_lambda) of suffix / prefix / indel-carrying reads against a full-length
    centre, band 32: lambda bits and hamming must equal the oracle's for every pair.  `wide` = k_nw_adw (run descriptors
    used to hold the diagonal offset in 8 bits), `lane` = k_nw<WMAX> / k_nw_gen."""
    monkeypatch.setenv("DADA2HIP_NW_KERNEL", nw_kernel)
    rng = np.random.default_rng(L)
    centre = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=L))
    seqs = [centre]
    for off in offsets:
        seqs.append(_mutate(rng, centre[off:], nsub=3))                                  # 5' truncation
        seqs.append(_mutate(rng, centre[off:], nsub=2, dels=(50, 51), ins=(200,)))       # ... with internal indels
        seqs.append(_mutate(rng, centre[off // 2: L - off // 2], nsub=4))                # both ends ragged
        seqs.append(_mutate(rng, centre[: L - off], nsub=1, ins=(30, 31, 32)))           # 3' truncation + insertion
    seqs = list(dict.fromkeys(seqs))
    n = len(seqs)
    maxlen = max(len(s) for s in seqs)
    quals = np.full((n, maxlen), np.nan)
    for i, s in enumerate(seqs):
        quals[i, : len(s)] = rng.integers(5, 41, size=len(s))
    ab = np.array([1000] + [5] * (n - 1), dtype=np.int32)
    err = tperr1()
    o = DadaOpts(BAND_SIZE=32)
    smp = api.Sample(seqs, ab, None, quals)
    try:
        lam, ham, cls, st = smp.compare(0, err, o, kdist_cutoff=1.0)
    finally:
        smp.close()
    for i in range(1, n):
        wl, wh, kd, ko = oracle_c.compare(seqs[0], quals[0, : len(seqs[0])], seqs[i], quals[i, : len(seqs[i])], err, o, kdist_cutoff=1.0)
        assert ham[i] == wh, (i, len(seqs[i]), int(ham[i]), wh)
        assert lam[i] == wl, (i, len(seqs[i]), lam[i], wl)


@pytest.mark.parametrize("nw_kernel", ["wide", "lane", "auto"])
def test_five_prime_ragged_sample_whole_path(api, oracle_c, nw_kernel, monkeypatch):
    """Whole dada_uniques on a long-read-shaped sample whose true variants are 5'-ragged (start offsets up to 400 nt) and
    carry internal indels at long-read rates."""
    if nw_kernel != "auto":
        monkeypatch.setenv("DADA2HIP_NW_KERNEL", nw_kernel)
    from dada2_amd.synth import make_sample
    d = make_sample(tperr1(), 350, L=620, G=16, seed=4242, Lmin5=220, indel_rate=2e-3, ins_rate=2e-3, chunk=3000)
    assert max(len(s) for s in d.seqs) - min(len(s) for s in d.seqs) > 300
    o = DadaOpts(BAND_SIZE=32)
    got = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, o)
    want = oracle_c.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, o)
    assert_results_equal(got, want, p_rtol=P_RTOL)


# ---- unsorted input: b_bud's "slot 0 is the centre" quirk depends on the slot order (cluster.cpp:285) -------------------
@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_randomly_permuted_input_matches_oracle(api, oracle_c, seed):
    from dada2_amd.synth import make_sample
    d = make_sample(tperr1(), 600, L=110, G=12, seed=500 + seed, chunk=4000)
    perm = np.random.default_rng(seed).permutation(d.nraw)
    seqs = [d.seqs[i] for i in perm]
    ab, q = d.abundances[perm].copy(), d.quals[perm].copy()
    o = DadaOpts(GREEDY=bool(seed & 1))
    got = api.dada_uniques(seqs, ab, None, tperr1(), q, o)
    want = oracle_c.dada_uniques(seqs, ab, None, tperr1(), q, o)
    assert_results_equal(got, want, p_rtol=P_RTOL)


# ---- near-tied bud candidates: the host must take the reference's decision with the CPU's arithmetic --------------------
def test_near_tied_bud_candidates_follow_cpu_order(api, oracle_c):
    """A flat error matrix and constant qualities make the lambdas of equally abundant Hamming-2 variants products of
    the SAME factors in a different order: their p-values agree to ~1e-15 without being bit-equal, and device libm
    differs from the host's in the last ulp.  b_bud's arg-min among them must still be the CPU reference's."""
    rng = np.random.default_rng(77)
    L = 120
    centre = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=L))
    seqs, ab = [centre], [4000]
    for k in range(14):
        seqs.append(_mutate(rng, centre, nsub=2))
        ab.append(60)
    for k in range(40):
        seqs.append(_mutate(rng, centre, nsub=1))
        ab.append(int(rng.integers(1, 4)))
    seqs = list(dict.fromkeys(seqs))
    ab = np.array(ab[: len(seqs)], dtype=np.int32)
    order = np.argsort(-ab, kind="stable")
    seqs, ab = [seqs[i] for i in order], ab[order]
    quals = np.full((len(seqs), L), 30.0)
    err = np.full((16, 41), 1e-3)
    err[[0, 5, 10, 15], :] = 1.0 - 3e-3
    for o in (DadaOpts(), DadaOpts(OMEGA_A=1e-10, MIN_HAMMING=2)):
        got = api.dada_uniques(seqs, ab, None, err, quals, o)
        want = oracle_c.dada_uniques(seqs, ab, None, err, quals, o)
        assert got.nclust == want.nclust >= 3
        assert_results_equal(got, want, p_rtol=P_RTOL)
        # the birth p-values are recomputed on the host with the reference's arithmetic: bit-equal, not just close
        assert np.array_equal(got.clustering["birth_pval"][1:], want.clustering["birth_pval"][1:])


# ---- exact ties of b_bud's best key: the first in partition order, then in member-list order (cluster.cpp:284-308) ---------
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_exact_bud_ties_follow_the_member_list_order(api, oracle_c, seed):
    """Equal-read candidates with p = 0: the device settles the ties whose order needs no member list (never-moved members of
    partition 0, a single candidate in the lowest partition), the host the others - all must be the reference's choice."""
    from helpers import zero_tie_sample
    seqs, ab, q = zero_tie_sample(seed)
    for o in (DadaOpts(), DadaOpts(OMEGA_A=1e-4, DETECT_SINGLETONS=True)):
        got = api.dada_uniques(seqs, ab, None, tperr1(), q, o)
        want = oracle_c.dada_uniques(seqs, ab, None, tperr1(), q, o)
        assert got.nclust == want.nclust >= 8
        assert_results_equal(got, want, p_rtol=P_RTOL)


# ---- batch C entry point (dada2hip_run_multi): one host thread per device entry ------------------------------------------
@pytest.mark.parametrize("slots", ["3", "1"], ids=["tails-side-by-side", "tails-take-turns"])
def test_run_multi_equals_per_sample_calls(api, oracle_c, monkeypatch, slots):
    from dada2_amd.synth import make_sample
    # (round 6: up to three samples hold a persistent slot of the device side by side; DADA2HIP_V3_SLOTS=1 = their rounds take turns)
    monkeypatch.setenv("DADA2HIP_V3_SLOTS", slots)
    dereps = [make_sample(tperr1(), 500 + 40 * i, L=100, G=8, seed=900 + i, chunk=3000) for i in range(5)]
    res = api.dada_uniques_multi(dereps, tperr1(), DadaOpts(), devices=(0, 0, 0))   # three host threads share the one GPU here
    assert len(res) == 5
    for d, r in zip(dereps, res):
        want = oracle_c.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts())
        assert_results_equal(r, want, p_rtol=P_RTOL)
    from dada2_amd import _lib
    bad = Derep(["ACGTNACGTA", "ACGTAACGTA"], np.array([5, 1], dtype=np.int32), np.full((2, 10), 30.0), np.zeros(0, dtype=np.int32))
    with pytest.raises(_lib.Dada2HipError, match="sample 2"):
        api.dada_uniques_multi([dereps[0], bad], tperr1(), DadaOpts(), devices=(0,))


def test_several_samples_on_one_gpu_take_the_same_time_run_after_run(tmp_path):
    """dada2hip_run_multi with four samples of 60 k uniques, two in flight on the one GPU, ten times in fresh
    processes - no run may take more than 1.5 x the median (round 5 saw a bimodal 234 / 574-1 250 ms on configs[3] before the
    prefetch gate's bound went from 500 to 5 ms: a waiting gate kernel at the head of a hardware queue the tail's stream shared),
    and every run returns the same partitions.  (The four samples are one seeded sample and three copies whose abundances differ:
    drawing a 60 k sample takes half a minute of numpy.)"""
    import json, pickle, subprocess, sys
    from dada2_amd.synth import make_sample
    d0 = make_sample(tperr1(), 60000, L=250, G=48, seed=5100, chunk=60000)
    with open(tmp_path / "sample.pkl", "wb") as fh:
        pickle.dump((d0.seqs, d0.abundances, d0.quals), fh, protocol=4)
    code = (
        "import sys, time, json, pickle\n"
        "import numpy as np\n"
        "root = %r\n"
        "sys.path[:0] = [root, root + '/tests']\n"
        "from helpers import tperr1\n"
        "from dada2_amd import api\n"
        "from dada2_amd.io import Derep\n"
        "from dada2_amd.opts import DadaOpts\n"
        "seqs, ab, q = pickle.load(open(%r, 'rb'))\n"
        "dereps = []\n"
        "for i in range(4):\n"
        "    a = ab.copy(); a[: 40 * i] += 1            # (still sorted: the first uniques are the abundant ones)\n"
        "    dereps.append(Derep(seqs, a, q, np.zeros(0, np.int32)))\n"
        "his = [api.HostInput.from_derep(d) for d in dereps]        # (the C-side buffers, outside the timed call: 60 k Python strings each)\n"
        "api.dada_uniques_multi(his[:2], tperr1(), DadaOpts(), devices=(0, 0))\n"      # (warm-up: allocation cache, attributes)
        "t0 = time.perf_counter()\n"
        "res = api.dada_uniques_multi(his, tperr1(), DadaOpts(), devices=(0, 0))\n"
        "print(json.dumps({'ms': (time.perf_counter() - t0) * 1e3, 'nclust': [int(r.nclust) for r in res],\n"
        "                  'rounds_ms': [float(r.stats['ms_bookkeep']) for r in res], 'upload_ms': [float(r.stats['ms_upload']) for r in res],\n"
        "                  'final_ms': [float(r.stats['ms_final']) for r in res],\n"
        "                  'detail': [{k: round(float(r.stats[k]), 1) for k in ('ms_bookkeep', 'ms_wait_device', 'ms_replay', 'ms_enqueue', 'ms_setup', 'ms_round0', 'tail_launches', 'tail_fallbacks', 'batch_compares', 'lite_misses', 'rounds')} for r in res]}))\n"
    ) % (ROOT, str(tmp_path / "sample.pkl"))
    def throttled_usec():
        # The GPU boxes are containers with a CPU QUOTA (cpu.max = 1 600 000 / 100 000: sixteen CPUs' worth per 100 ms, on a host that
        # shows 256 cores), and the kernel throttles them in more than half of all periods while this suite runs (cpu.stat:
        # nr_throttled 1 667 of 2 979 periods over two runs of this test).  When the quota of a period is used up EVERY thread of
        # the container stands still until the period ends - up to 100 ms in which a sample's upload, its replay, its setup and
        # its wait for the device all "take" 80-100 ms at once.  throttled_usec is summed over the CPUs, so it is never zero here
        # (5-9 s per child of this test); with another CPU-heavy process in the container - a reference worker of the at-size
        # cases - it reads 20-600 s.  A run whose figure is more than three times the median of the runs was disturbed from
        # outside and is not judged; the rest are.  
        try:
            for line in open("/sys/fs/cgroup/cpu.stat"):
                if line.startswith("throttled_usec"):
                    return int(line.split()[1])
        except OSError:
            pass
        return 0

    runs = []
    for k in range(14):
        t_thr = throttled_usec()
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
        r = json.loads(out.stdout.strip().splitlines()[-1])
        r["throttled_ms"] = (throttled_usec() - t_thr) / 1e3
        runs.append(r)
        thr_med = sorted(x["throttled_ms"] for x in runs)[len(runs) // 2]
        if len(runs) >= 10 and sum(1 for x in runs if x["throttled_ms"] <= 3.0 * thr_med) >= 10:
            break
    thr_med = sorted(x["throttled_ms"] for x in runs)[len(runs) // 2]
    print("container CPU throttling during the runs (ms, summed over CPUs):", [round(r["throttled_ms"]) for r in runs])
    judged = [r for r in runs if r["throttled_ms"] <= 3.0 * thr_med]
    all_runs, runs = runs, judged
    ms = sorted(r["ms"] for r in runs)
    ms += [ms[-1]] * (10 - len(ms))                              # (the criteria below index ten runs)
    med = 0.5 * (ms[4] + ms[5])
    print("four samples of 60 k uniques, two in flight, ten processes: ms", [round(m, 1) for m in ms], runs[0]["nclust"])
    assert all(r["nclust"] == runs[0]["nclust"] for r in runs), [r["nclust"] for r in runs]
    assert min(runs[0]["nclust"]) > 20
    # What is guarded: a stall of the ROUNDS - round 5's was one sample waiting two bounds of a prefetch gate for a result block
    # (574-1 250 ms against 234).  The library's own clock of every sample's rounds (ms_bookkeep: from the first persistent launch
    # to the last result block) must stay within 3 x its median over the 40 samples of the ten runs (two exceptions, below).  The wall of a whole call
    # is printed and only loosely bounded: single runs are held up on the HOST side now and then (one sample's upload or final
    # pass 20 ms instead of 3: host pool, allocations - tools/multi_stall.py shows where; 48 fresh processes outside pytest read
    # 28.7-41.0 ms, profiles/r09w_four_samples_two_in_flight_*.jsonl, inside a pytest session 56-254 ms have been seen against a
    # median of 31), which is not what this test is about.
    rounds = sorted(x for r in runs for x in r["rounds_ms"])
    rmed = rounds[len(rounds) // 2]
    for r in runs:
        if max(r["rounds_ms"]) > 3.0 * rmed:
            print("a run with slow rounds: wall %.1f ms, per sample:" % r["ms"], r["detail"])
    print("rounds per sample (library clock): median %.1f ms, max %.1f ms; slowest run's upload / final ms:" % (rmed, rounds[-1]),
          [round(x, 1) for x in max(runs, key=lambda r: r["ms"])["upload_ms"]], [round(x, 1) for x in max(runs, key=lambda r: r["ms"])["final_ms"]])
    # (round 5's stall was 40 x the median)
    # (... one sample in ten: the container's own throttling - see throttled_usec above - freezes a pair of samples for the rest of a
    #  100 ms period now and then, also without a disturber: 39 / 39, 60 / 61 / 90 / 90 ms have been seen in full runs of the suite)
    assert rounds[-(len(rounds) // 10) - 1] <= 3.0 * rmed and rounds[-1] <= 30.0 * rmed, ("samples' rounds stalled", [round(x, 1) for x in rounds[-8:]], rmed)
    assert ms[5] <= 1.6 * ms[1] and ms[-1] <= 20.0 * med, ("whole calls stalled", [round(m, 1) for m in ms])   # (not bimodal; no run out of all proportion)


def test_nwalign_short_strings_and_limits(api):
    """C_nwalign / C_nwvec accept any length (evaluate.cpp:18); the device helper path has no k-mer-size limit."""
    from oracle import cport
    for a, b in (("ACG", "AG"), ("A", "A"), ("ACGTT", "ACGT"), ("TTACGTACGTAA", "ACGTACGT")):
        assert api.nwalign(a, b, band=-1) == cport.nwalign(a, b, band=-1), (a, b)
    from dada2_amd import _lib
    with pytest.raises(_lib.Dada2HipError) as ei:
        api.nwalign("ACGTN", "ACGTA", band=-1)
    assert ei.value.code == 4 and "A/C/G/T" in str(ei.value)


def test_use_quals_off_still_checks_the_err_range(api):
    """The output tables index err's columns by quality whatever USE_QUALS says: narrower err -> the reference's error."""
    from dada2_amd import _lib
    from dada2_amd.synth import make_sample
    d = make_sample(tperr1(), 100, L=60, G=4, seed=302, chunk=2000)
    co = DadaOpts().to_c()
    co.use_quals = 0
    with pytest.raises(_lib.Dada2HipError, match="exceeded range of err lookup table"):
        api.dada_uniques(d.seqs, d.abundances, None, tperr1()[:, :20], d.quals, copts=co)


# ---- the two round engines: classic host-stepped rounds vs batched compares + device-driven rounds ----------
def _run_cases_in_subprocess(env_extra, names, seeded=()):
    import subprocess, sys
    code = (
        "import numpy as np, sys\n"
        "root = %r\n"
        "sys.path[:0] = [root, root + '/tests']\n"
        "from helpers import case_inputs, assert_results_equal, P_RTOL, tperr1\n"
        "from dada2_amd import api\n"
        "from dada2_amd.opts import DadaOpts\n"
        "from dada2_amd.synth import make_sample\n"
        "from oracle import cport\n"
        "for name in %r:\n"
        "    d, err, pri, o, exp, meta = case_inputs(name)\n"
        "    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)\n"
        "    assert_results_equal(got, exp, p_rtol=P_RTOL, check_birth_from=pri is None)\n"
        "for seed, n, L, G in %r:\n"
        "    d = make_sample(tperr1(), n, L=L, G=G, seed=seed, chunk=20000)\n"
        "    got = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts())\n"
        "    want = cport.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts())\n"
        "    assert_results_equal(got, want, p_rtol=P_RTOL)\n"
        "    assert got.stats['ncompare'] - got.stats['nskipped'] == want.stats['nalign'] and got.stats['nshroud'] == want.stats['nshroud']\n"
        "print('ok')\n") % (ROOT, tuple(names), tuple(seeded))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


_ALL = ("sam1F_default", "sam1F_band32_omega20_inflate3", "sam1R_default", "sam2F_nogreedy", "sam1F_nokmers", "sam2R_singletons",
        "sam1F_priors", "synth3000_default")
_SEEDED = ((7001, 6000, 200, 48), (7002, 20000, 250, 96))


@pytest.mark.parametrize("env", [
    {"DADA2HIP_ENGINE": "classic"},                       # round-1 loop
    {},                                                   # v2 defaults: 8 batch buffers x 8 centres, 2 chains in flight, 4 shuffles per chain
    {"DADA2HIP_V2_NBUF": "1"},                            # one batch buffer: every miss evicts everything
    {"DADA2HIP_V2_DEPTH": "1"},                           # host in lockstep with the device
    {"DADA2HIP_V2_DEPTH": "3"},
    {"DADA2HIP_V2_CHAIN": "1"},                           # one shuffle per chain: rounds continue through the host (H2_SHUFFLE_MORE)
    {"DADA2HIP_V2_CHAIN": "2", "DADA2HIP_NODE_CAP": "1"}, # comparison store starts at N + 16 blocks: growth through H2_CAPACITY
    {"DADA2HIP_V2_ALIGN": "commit"},                      # each centre's pairs aligned when its round commits (the long-read mode)
    {"DADA2HIP_V2_LITE": "0"},                            # every chain carries the batch compare's launches (no H2_NEED_COMPARE)
    {"DADA2HIP_V2_GRAPH": "1", "DADA2HIP_V2_NBUF": "2"},  # hipGraph replay of both chain forms, frequent evictions
    # round 4: the persistent tail is the default above; here with several blocks on small samples, with pauses (movers that
    # do not fit the result block), a lagging host (ring limit), a tiny product buffer (in-kernel lambda for the overflow) -
    # and the launch chains, which the settings above that name chain knobs no longer reach by themselves
    {"DADA2HIP_V3_GRID": "7", "DADA2HIP_V2_MOV_INLINE": "64", "DADA2HIP_V3_RING": "2", "DADA2HIP_AD_FCAP": "1000"},
    {"DADA2HIP_V2_TAIL": "chain"},
    {"DADA2HIP_V2_TAIL": "chain", "DADA2HIP_V2_CHAIN": "1", "DADA2HIP_V2_MOV_INLINE": "64"},
    {"DADA2HIP_V2_TAIL": "chain", "DADA2HIP_V2_LITE": "0", "DADA2HIP_V2_ALIGN": "commit"},
    # round 5: the next batch's compare runs under the tail by default (the "v2" line above); the same rounds without it,
    # with compares the HOST launches when it sees the plan (no chain ahead behind a gate), with every prefetch waited for at
    # the next serial end, and with a tail that leaves the launch at once instead of spinning for a prefetch that is late
    {"DADA2HIP_V3_OVERLAP": "0"},
    {"DADA2HIP_V3_PF_GATE_US": "0"},
    {"DADA2HIP_V3_PF_SYNC": "1", "DADA2HIP_V3_GRID": "5"},
    {"DADA2HIP_V3_PF_WAIT_US": "0", "DADA2HIP_V3_PF_EARLY": "0", "DADA2HIP_V2_NBUF": "4"},
    # ... the round's evaluation on every shuffle call behind the commit's (void attempts: rewritten p-values, locks taken back, the
    # candidate list emptied), and as a phase of its own (round 4's form); the default attempts behind calls that moved <= 16 uniques
    {"DADA2HIP_V3_SPEC_MAX": "1000000000"},
    {"DADA2HIP_V3_SPEC_MAX": "1000000000", "DADA2HIP_V3_GRID": "6", "DADA2HIP_V3_OVERLAP": "0"},
    {"DADA2HIP_V3_SPEC": "0"},
    # ... and attempts only behind calls that moved <= 2 / <= 3 uniques: attempts that stand, attempts that are void and PLAIN calls
    # mix within one round (a lock decided by a void attempt must never be seen by anybody - the locks of an attempt are
    # published only once it stands, Eng2::spec_lock_buf)
    {"DADA2HIP_V3_SPEC_MAX": "2"},
    # round 6: the batch screen without its 5-mer presence bitmaps (every unique through the exact k-mer walk) and the batch aligner
    # without its pointer-free first pass (every pair through the full kernel) - what the defaults above must agree with
    {"DADA2HIP_SCREEN_BITS": "0", "DADA2HIP_AD_FAST": "0"},
    {"DADA2HIP_SCREEN_BITS": "0", "DADA2HIP_V2_TAIL": "chain"},
    {"DADA2HIP_V3_BLOCK": "512"},                         # round 5's tail under the overlap: 512-thread blocks beside the compares on every CU
    {"DADA2HIP_V3_SPEC_MAX": "3", "DADA2HIP_V3_GRID": "5", "DADA2HIP_V3_PF_EARLY": "0"},
    # the XCD-hierarchical grid barrier (default from 48 blocks on) forced onto small grids, and the flat one forced onto the defaults
    {"DADA2HIP_V3_XBAR": "1", "DADA2HIP_V3_GRID": "9"},
    {"DADA2HIP_V3_XBAR": "1", "DADA2HIP_V3_GRID": "64", "DADA2HIP_V2_MOV_INLINE": "64"},
    {"DADA2HIP_V3_XBAR": "0"},
    # round 6: the tail's LDS mirror of the per-unique facts its sweeps ask for (default: on) - off, and on with the check that
    # compares it with the state arrays at the end of every round (several blocks; attempts, void attempts and plain calls mixed;
    # 512-thread blocks; pauses and launches left for prefetches: the mirror is refilled at every launch entry)
    {"DADA2HIP_V3_MIRROR": "0"},
    {"DADA2HIP_V3_MIRROR": "2"},
    {"DADA2HIP_V3_MIRROR": "2", "DADA2HIP_V3_GRID": "5", "DADA2HIP_V3_SPEC_MAX": "3", "DADA2HIP_V2_MOV_INLINE": "64"},
    {"DADA2HIP_V3_MIRROR": "2", "DADA2HIP_V3_GRID": "3", "DADA2HIP_V3_BLOCK": "512", "DADA2HIP_V3_PF_WAIT_US": "0", "DADA2HIP_V2_NBUF": "4"},
    # ... the host's replay of the published moves and births on a second host thread (the replay lane: not the default), with
    # halts that hand the mirror back to the boundary thread (host decisions, pauses, growth); and two persistent launches in flight
    {"DADA2HIP_V3_LANE": "1"},
    {"DADA2HIP_V3_LANE": "1", "DADA2HIP_V3_GRID": "4", "DADA2HIP_V2_MOV_INLINE": "64", "DADA2HIP_NODE_CAP": "1", "DADA2HIP_V3_RING": "2"},
    {"DADA2HIP_V2_DEPTH": "2"},
    # the host's replay orders EVERY mover list with its radix sort (by default only lists of 4 096 movers and more)
    {"DADA2HIP_REPLAY_RADIX_MIN": "1"},
], ids=["classic", "v2", "v2-nbuf1", "v2-depth1", "v2-depth3", "v2-chain1", "v2-chain2-grow", "v2-align-commit", "v2-nolite", "v2-graph",
        "tail-grid7-pauses-ring2-fcap", "chains", "chains-chain1-biglists", "chains-nolite-commit",
        "tail-serial", "overlap-host-launched", "overlap-sync-grid5", "overlap-leave-at-once-nbuf4",
        "evaluate-on-every-call", "evaluate-on-every-call-grid6-serial", "evaluate-apart", "attempts-and-plain-calls-mixed", "exact-screen-full-aligner", "exact-screen-chains", "tail-512-thread-blocks",
        "attempts-and-plain-calls-mixed-grid5",
        "xcd-barrier-grid9", "xcd-barrier-grid64-pauses", "flat-barrier",
        "no-mirror", "mirror-checked", "mirror-checked-grid5-mixed-calls-pauses", "mirror-checked-grid3-512-leaves-for-prefetches",
        "replay-lane", "replay-lane-grid4-pauses-growth-ring2", "two-launches-in-flight", "replay-radix-sort-always"])
def test_round_engines_agree_with_the_reference(env):
    """Every engine configuration must reproduce the goldens the reference produced, the oracle on two seeded samples (6 k and
    20 k uniques: dozens of rounds, multi-shuffle rounds, cache hits and misses) and the reference's own work counters."""
    _run_cases_in_subprocess(env, _ALL, _SEEDED)


@pytest.mark.parametrize("L,band,n", [(250, 16, 30000), (100, 16, 8000), (251, 8, 8000), (333, 18, 6000), (120, 1, 4000)])
def test_equal_length_samples_of_several_lengths_and_bands_match_oracle(L, band, n):
    """Equal-length samples (the per-round aligner's steady state covers the whole matrix interior) of several lengths and
    bands - odd lengths, a band of one cell - against the C restatement."""
    import subprocess, sys
    code = (
        "import sys\n"
        "root = %r\n"
        "sys.path[:0] = [root, root + '/tests']\n"
        "from helpers import assert_results_equal, P_RTOL, tperr1\n"
        "from dada2_amd import api\n"
        "from dada2_amd.opts import DadaOpts\n"
        "from dada2_amd.synth import make_sample\n"
        "from oracle import cport\n"
        "import numpy as np\n"
        "from dada2_amd.io import Derep\n"
        "d0 = make_sample(tperr1(), %d, L=%d + 3, G=64, seed=%d, chunk=20000, indel_rate=0.0006, ins_rate=0.0006)\n"
        "Lt = %d\n"   # truncLen: reads with an insertion / deletion keep their shifted tail, all uniques end up Lt long
        "first, ab = {}, {}\n"
        "for k, sq in enumerate(d0.seqs):\n"
        "    if len(sq) < Lt: continue\n"
        "    t = sq[:Lt]\n"
        "    first.setdefault(t, k); ab[t] = ab.get(t, 0) + int(d0.abundances[k])\n"
        "keys = sorted(ab, key=lambda t: (-ab[t], first[t]))\n"
        "d = Derep(keys, np.array([ab[t] for t in keys], dtype=np.int32), np.stack([d0.quals[first[t], :Lt] for t in keys]), np.zeros(0, np.int32))\n"
        "assert len(set(map(len, d.seqs))) == 1 and d.abundances[0] >= d.abundances[1]\n"
        "o = DadaOpts(BAND_SIZE=%d)\n"
        "got = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, o)\n"
        "want = cport.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, o)\n"
        "assert_results_equal(got, want, p_rtol=P_RTOL)\n"
        "assert got.nclust > 5 and got.stats['nnw'] > 100\n"
        "print('ok', got.nclust, got.stats['nnw'])\n") % (ROOT, n, L, 9000 + L + band, L, band)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


# ---- the RCCL path of the sample-sharded driver, real resident-sample runner, world_size 1 -------------------------------
def test_dada_multi_with_real_runner_under_nccl(api):
    import socket
    import torch
    import torch.distributed as dist
    from dada2_amd.multi import dada_multi
    from dada2_amd.synth import make_sample
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dereps = [make_sample(tperr1(), 400, L=100, G=8, seed=100 + i, chunk=2000) for i in range(3)]
        o = DadaOpts(OMEGA_C=0, MAX_CONSIST=3)
        res, err, errs = dada_multi(dereps, None, self_consist=True, opts=o, dist=dist, device=torch.device("cuda", 0))
        res1, err1, errs1 = api.dada(dereps, None, self_consist=True, opts=o)
        assert np.array_equal(err, err1) and len(errs) == len(errs1)
        for i in range(3):
            assert_results_equal(res[i], res1[i], exact_float=True)
    finally:
        dist.destroy_process_group()


# ---- BASELINE.json's sizes ------------------------------------------------------------------------------------------------
def test_headline_1M_uniques_full_parity_vs_reference_itself(api):
    """BASELINE.json's headline size: 1 000 000 unique 250-nt reads (bench.py's default workload, same seed).  EVERY output
    of the GPU run against the reference's own C++ (about a minute of its multithreaded path, run in the background since the
    session started: tests/at_size.py).  Skipped without oracle/_ref."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not present")
    import at_size
    dereps, err, o, wants = at_size.get("cfg3")
    d, want = dereps[0], wants[0]
    assert d.nraw == 1_000_000
    got = api.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    assert got.nclust == want.nclust > 300
    assert_results_equal(got, want, p_rtol=P_RTOL)


def test_config5_long_reads_parity_vs_reference_itself(api):
    """BASELINE.json configs[4] shape: ~1 500-nt reads (1 450..1 510, 3'-ragged), band 32, 94 quality columns, indels; sized
    (50 k uniques, MAX_CLUST 16) so that the all-core reference finishes in about a minute."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not present")
    from dada2_amd.synth import make_sample
    err = extend_err(tperr1(), 93)
    d = make_sample(err, 50_000, L=1510, G=128, seed=20260925 + 5, Lmin=1450, q_hi=93.0, q_lo=30.0, q_max=93, indel_rate=1e-4,
                    ins_rate=1e-4, chunk=10_000)
    o = DadaOpts(BAND_SIZE=32, MAX_CLUST=16)
    got = api.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    ref.set_threads(os.cpu_count() or 1)
    want = ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o, multithread=True)
    ref.set_threads(1)
    assert got.nclust == want.nclust == 16
    assert_results_equal(got, want, p_rtol=P_RTOL)


# ---- bimera identification: the step after dada() ---------------------------------------------------
def test_bimera_table_matches_reference_goldens(api):
    z = np.load(os.path.join(ROOT, "tests", "golden", "bimera_table.npz"))
    mat, seqs = z["mat"], [str(s) for s in z["seqs"]]
    for oo in (0, 1):
        for ms in (16, 4):
            nflag, nsam = api.table_bimera2(mat, seqs, allow_one_off=bool(oo), max_shift=ms)
            assert np.array_equal(nflag, z[f"nflag_oo{oo}_ms{ms}"]), (oo, ms, nflag.tolist())
            assert np.array_equal(nsam, z[f"nsam_oo{oo}_ms{ms}"])
        tot = mat.sum(axis=0)
        for j, s in enumerate(seqs):
            pars = [seqs[k] for k in range(len(seqs)) if tot[k] > 2 * tot[j] and tot[k] > 8]
            assert api.is_bimera(s, pars, allow_one_off=bool(oo)) == bool(z[f"isbim_oo{oo}"][j]), (j, oo)


@pytest.mark.parametrize("seed,nseq,nsam,L", [(1, 60, 3, 120), (2, 300, 6, 250), (3, 120, 2, 400)])
def test_bimera_table_seeded_vs_oracle(api, oracle_c, seed, nseq, nsam, L):
    """Random tables: true sequences at a few % divergence, bimeras / one-offs / shifted / indel-carrying mosaics of them at low
    abundance, ragged lengths; every (sequence, parent) decision against the oracle."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, size=L)
    true = []
    for g in range(max(6, nseq // 5)):
        s = anc.copy()
        p = rng.choice(L, size=int(L * rng.uniform(0.03, 0.12)), replace=False)
        s[p] = (s[p] + rng.integers(1, 4, size=p.size)) & 3
        true.append("".join("ACGT"[x] for x in s[: L - int(rng.integers(0, 6))]))
    seqs = list(dict.fromkeys(true))
    while len(seqs) < nseq:
        a, b = rng.choice(len(true), 2, replace=False)
        cut = int(rng.integers(10, L - 10))
        ch = true[a][:cut] + true[b][cut:]
        kind = int(rng.integers(0, 5))
        if kind == 1:
            ch = _mutate(rng, ch, nsub=1)
        elif kind == 2:
            ch = ch[int(rng.integers(1, 20)):]
        elif kind == 3:
            ch = _mutate(rng, ch, dels=(min(len(ch) - 2, cut + 3),))
        elif kind == 4:
            ch = _mutate(rng, ch, nsub=3)
        if ch not in seqs:
            seqs.append(ch)
    mat = np.zeros((nsam, len(seqs)), dtype=np.int32)
    nt = len(dict.fromkeys(true))
    mat[:, :nt] = rng.integers(0, 400, size=(nsam, nt)) * (rng.random((nsam, nt)) < 0.8)
    mat[:, nt:] = rng.integers(0, 12, size=(nsam, len(seqs) - nt)) * (rng.random((nsam, len(seqs) - nt)) < 0.6)
    for oo, mf, ma, ms in ((False, 1.5, 2, 16), (True, 1.5, 2, 16), (True, 2.0, 8, 8)):
        got = api.table_bimera2(mat, seqs, min_fold=mf, min_abund=ma, allow_one_off=oo, max_shift=ms)
        want = oracle_c.table_bimera2(mat, seqs, min_fold=mf, min_abund=ma, allow_one_off=oo, max_shift=ms)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (oo, mf, ma, ms)
    flagged = api.is_bimera_denovo_table(mat, seqs)
    assert flagged.dtype == bool and flagged.shape == (len(seqs),)


@pytest.mark.parametrize("nw_kernel", ["coop", "lane"])
def test_bimera_pair_quantities_match_the_reference(api, oracle_c, nw_kernel, monkeypatch):
    """What C_is_bimera / C_table_bimera2 look at, per alignment (get_lr / get_ham_endsfree, chimera.cpp:211-293): k_nw_ad in its
    bimera mode (coop: the default wherever its geometry applies) and the lane kernel + k_bimera_lr, against the oracle - and the
    reference's own functions where `_ref` is present (test_oracle.py pins the one to the other)."""
    from helpers import BIMERA_PAIR_OPTIONS, bimera_pair_cases
    from oracle import ref
    monkeypatch.setenv("DADA2HIP_NW_KERNEL", nw_kernel)
    for seed, n, L in ((1, 400, 60), (2, 400, 130), (3, 300, 251), (4, 120, 600)):
        qs, ps = bimera_pair_cases(seed, n, L)
        for oo, ms, sc in BIMERA_PAIR_OPTIONS:
            want = oracle_c.bimera_pairs(qs, ps, oo, *sc, ms)
            got = api.bimera_pairs(qs, ps, oo, *sc, ms)
            assert np.array_equal(got, want), (seed, oo, ms, sc, np.nonzero((got != want).any(axis=1))[0][:5])
        if ref.available():
            assert np.array_equal(api.bimera_pairs(qs, ps, True), ref.bimera_pairs(qs, ps, True))
    from helpers import bimera_short_pair_cases
    qs, ps = bimera_short_pair_cases(5, 600)            # shorter than the band, all-gap alignments, maxShift 1 .. 40
    for oo, ms in ((True, 16), (False, 16), (True, 4), (True, 1), (True, 40)):
        assert np.array_equal(api.bimera_pairs(qs, ps, oo, max_shift=ms), oracle_c.bimera_pairs(qs, ps, oo, max_shift=ms)), (oo, ms)


def test_bimera_pair_quantities_many_pairs_one_launch(api, oracle_c):
    """60 000 pairs of 250 nt in one call: more chunks than the launch has waves (every block walks several work slices and
    reuses its slot of the pointer ring), every pair against the oracle."""
    from helpers import bimera_pair_cases
    qs, ps = bimera_pair_cases(11, 60000, 250)
    got = api.bimera_pairs(qs, ps, True)
    want = oracle_c.bimera_pairs(qs, ps, True)
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0][:5]


def test_library_first_then_torch_share_one_hip_runtime():
    """the library used to need `import torch` BEFORE it (two HIP runtimes otherwise, torch then sees no device).
    dada2_amd._lib maps torch's bundled runtime first when torch is installed, so either order works: a fresh interpreter that
    calls the library first and imports torch afterwards must still see the GPU through torch, and both must keep working."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from dada2_amd import api\n"
        "assert 'torch' not in sys.modules\n"
        "a = api.nwalign('ACGTTACGTAACGT', 'ACGTACGTAACGT', band=-1)\n"
        "import torch\n"
        "assert torch.cuda.is_available(), 'torch lost the device'\n"
        "x = torch.arange(8, device='cuda').sum().item()\n"
        "b = api.nwalign('ACGTTACGTAACGT', 'ACGTACGTAACGT', band=-1)\n"
        "assert a == b and x == 28\n"
        "print('ok')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-1000:] + out.stderr[-3000:]


# ---- the boundary's callbacks and the default engine's way out ------------------------------------------------
ENGINE_ENVS = [{}, {"DADA2HIP_V3_GRID": "3"}, {"DADA2HIP_V2_TAIL": "chain"}, {"DADA2HIP_ENGINE": "classic"}]
ENGINE_IDS = ["persistent-tail", "persistent-tail-grid3", "chains", "classic-engine"]


@pytest.mark.parametrize("env", ENGINE_ENVS, ids=ENGINE_IDS)
def test_abort_hook_then_clean_run_and_verbose_log(api, env, monkeypatch):
    """dada2hip_hooks: should_abort (= Rcpp::checkUserInterrupt, src/Rmain.cpp:330) ends a run at its 1st / 3rd / 8th round with
    DADA2HIP_ERR_ABORTED while launches of that run are still queued (the persistent slot held); the NEXT run in the same process
    equals the golden.  log (= the verbose Rprintfs, Rmain.cpp:317,333) carries one line per birth and the reference's counters."""
    from helpers import case_inputs
    from dada2_amd import _lib
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    d, err, pri, o, exp, meta = case_inputs("sam1F_default")
    for k in (1, 3, 8):
        polls = []

        def stop():
            polls.append(1)
            return len(polls) >= k
        with pytest.raises(_lib.Dada2HipError) as ei:
            api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o, should_abort=stop)
        assert ei.value.code == 5 and "aborted" in str(ei.value)
        assert len(polls) == k
        assert_results_equal(api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o), exp)
    lines = []
    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o, verbose=True, log=lines.append, should_abort=lambda: False)
    assert_results_equal(got, exp)
    text = "".join(lines)
    assert text.count("New Cluster C") == got.nclust - 1
    assert "ALIGN: 8655 aligns, 3032 shrouded (%d raw)." % len(d.seqs) in text


@pytest.mark.parametrize("nth", [1, 2])
def test_injected_entry_barrier_failure_continues_on_the_launch_chains(api, nth, monkeypatch):
    """(DADA2HIP_V3_FAIL_ENTRY=n: the n-th persistent launch gives up at its entry barrier, as one whose blocks cannot all be
    resident does; a run has at least two launches - the one behind round 0 and the one behind the first batch compare.)"""
    from helpers import case_inputs
    monkeypatch.setenv("DADA2HIP_V3_FAIL_ENTRY", str(nth))
    monkeypatch.setenv("DADA2HIP_V3_GRID", "7")
    d, err, pri, o, exp, meta = case_inputs("sam1F_default")
    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)
    assert_results_equal(got, exp)
    assert got.stats["tail_fallbacks"] == 1


def test_another_tenant_on_the_gpu_sends_the_run_to_the_launch_chains(api, monkeypatch):
    """k3_tail needs all its blocks resident at once.  With another tenant holding half of the CUs (tests/glue/occupy.hip: 128
    blocks on a stream of their own that each claim a CU's whole LDS for a few seconds) a 250-block launch cannot be: its entry
    barrier gives up after its 2-second bound with nothing changed, and the run continues on the launch chains - the result is
    the golden, not DADA2HIP_ERR_DEVICE."""
    import ctypes
    import time
    from helpers import case_inputs
    so = os.path.join(ROOT, "tests", "glue", "liboccupy.so")
    if not os.path.exists(so):
        pytest.skip("tests/glue/liboccupy.so not built")
    occ = ctypes.CDLL(so)
    occ.occupy_start.argtypes = [ctypes.c_int, ctypes.c_double]
    d, err, pri, o, exp, meta = case_inputs("sam1F_default")
    assert_results_equal(api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o), exp)   # (warm: allocations cached)
    monkeypatch.setenv("DADA2HIP_V3_GRID", "250")
    # 200 of the 256 CUs held: two 512-thread blocks of the tail fit a free CU, so at most 112 of the 250 become resident
    assert occ.occupy_start(200, 6000.0) == 0
    try:
        time.sleep(0.3)                                                    # (its blocks are resident)
        t0 = time.time()
        got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)
        dt = time.time() - t0
    finally:
        assert occ.occupy_wait() == 0
    assert_results_equal(got, exp)
    if got.stats["tail_fallbacks"] == 0 and dt > 4.0:
        # the run's launches never ran BESIDE the other tenant: the HIP runtime had put the two streams on one hardware queue
        # (it has a handful and deals streams over them), so the run simply waited the tenant out - nothing to fall back from
        pytest.skip("the tenant's stream and the library's shared a hardware queue (%.1f s): no concurrent launch to test" % dt)
    assert got.stats["tail_fallbacks"] == 1, (got.stats["tail_fallbacks"], dt)
