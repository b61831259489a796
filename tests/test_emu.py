"""The product's OWN kernels and driver, executed lane by lane on the CPU by the functional emulator under tests/emu
(TEST INFRASTRUCTURE: fibers per GPU thread, the cross-lane operations and barriers as rendezvous; see
tests/emu/hip/hip_runtime.h).  This is how kernel logic is debugged in the authoring container, which has no GPU; the
-m gpu tests remain the parity tests proper.  The emulated library is opened explicitly here (a subprocess that points
dada2_amd._lib at tests/emu/build/libdada2hip_emu.so) - nothing in dada2_amd/ knows it exists."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

# (the gate of a prefetch chain sent ahead of its plan always gives up under the emulator - launches run when they are enqueued, so the
#  plan is never there yet - and the chain is sent again when the plan shows up: a short bound keeps that from costing 0.7 s a time)
os.environ.setdefault("DADA2HIP_V3_PF_GATE_US", "20000")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
pytestmark = pytest.mark.skipif(not (os.path.exists(CXX) or shutil.which(CXX)), reason="no host clang++ for the emulator build")


# ---- the emulator jobs of this module run SIDE BY SIDE --------------------------------------------------------------------
# Every test here is one subprocess (the emulated library is opened in a process of its own) and the emulator is slow: one after
# the other they were seven of the CPU suite's ten minutes.  When the first test asks for the library, every collected test of the
# module that needs nothing but the library is called once "dry" - its body builds the command, _run records it and stops the
# body - and the recorded commands are started on a small pool; the tests then find their result waiting.  A command that was
# not recorded (a test with other fixtures, a second command of a test) simply runs when it is asked for.
class _DryStop(Exception):
    pass


_DRY = None          # a list while test bodies are being called dry
_POOL = None
_FUTS = {}


def _job_key(argv, env):
    e = env if env is not None else os.environ
    return (tuple(argv), tuple(sorted((k, v) for k, v in e.items() if k.startswith(("DADA2HIP_", "EMU_")))))


def _spawn(argv, env, timeout):
    return subprocess.run(argv, env=env, capture_output=True, text=True, timeout=timeout)


def _run(argv, env=None, capture_output=True, text=True, timeout=900):
    global _POOL
    if _DRY is not None:
        _DRY.append((list(argv), None if env is None else dict(env), timeout))
        raise _DryStop()
    k = _job_key(argv, env)
    if _POOL is None or k not in _FUTS:
        return _spawn(argv, env, timeout)
    return _FUTS.pop(k).result()


def _prefetch(items, lib):
    global _DRY, _POOL
    if _POOL is not None or os.environ.get("EMU_SERIAL"):
        return
    me = sys.modules[__name__]
    jobs = []
    for it in items:
        if getattr(it, "module", None) is not me or not hasattr(it, "obj"):
            continue
        params = dict(it.callspec.params) if hasattr(it, "callspec") else {}
        if not set(it.fixturenames) - {"request"} <= {"emu_lib"} | set(params):
            continue                                  # (tmp_path, the reference, ...: on demand)
        kw = dict(params)
        if "emu_lib" in it.fixturenames:
            kw["emu_lib"] = lib
        _DRY = []
        try:
            it.obj(**kw)
        except _DryStop:
            pass
        except Exception:                             # noqa: BLE001  (whatever it is, the real call will show it)
            pass
        jobs += _DRY
        _DRY = None
    if len(jobs) < 2:
        return
    from concurrent.futures import ThreadPoolExecutor
    _POOL = ThreadPoolExecutor(max_workers=max(2, min(6, (os.cpu_count() or 2) // 2)))
    for argv, env, timeout in jobs:
        k = _job_key(argv, env)
        if k not in _FUTS:
            _FUTS[k] = _POOL.submit(_spawn, argv, env, timeout)


@pytest.fixture(scope="module")
def emu_lib(request):
    import build as emu_build
    lib = emu_build.build()
    _prefetch(request.session.items, lib)
    return lib


def run_cases(emu_lib, names, env=None, timeout=900):
    code = (
        "import sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from dada2_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "from helpers import case_inputs, assert_results_equal\n"
        "from dada2_amd import api\n"
        "from oracle import cport\n"
        "for name in %r:\n"
        "    d, err, pri, o, exp, meta = case_inputs(name)\n"
        "    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)\n"
        "    assert_results_equal(got, exp, check_birth_from=pri is None)\n"
        "    st = got.stats\n"
        "    print('ok', name, got.nclust, st['nnw'], st['ngapless'], st['nshroud'], 'xcd-barrier' if st['tail_xcd_barrier'] else 'flat-barrier')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), emu_lib, tuple(names))
    e = dict(os.environ)
    e.update(env or {})
    out = _run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and out.stdout.count("ok ") == len(names), out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


@pytest.mark.parametrize("env", [{}, {"DADA2HIP_ENGINE": "classic"}, {"DADA2HIP_NW_KERNEL": "lane"}, {"DADA2HIP_NW_KERNEL": "wide"},
                                 {"DADA2HIP_V2_ALIGN": "commit", "DADA2HIP_V3_GRID": "2"},
                                 # the persistent round tail (k3_tail) with several co-resident blocks, with mover lists that do not
                                 # fit the result block (pause), with a host that lags (ring limit), growing its buffers
                                 {"DADA2HIP_V3_GRID": "3"}, {"DADA2HIP_V3_GRID": "5", "DADA2HIP_V2_MOV_INLINE": "8"},
                                 {"DADA2HIP_V3_GRID": "2", "DADA2HIP_V3_RING": "1", "DADA2HIP_V2_NBUF": "1", "DADA2HIP_NODE_CAP": "1",
                                  "DADA2HIP_AD_FCAP": "40"},   # (+ a product buffer of 40 rows: the rest is multiplied up inside k_nw_ad)
                                 # the serial form of the persistent tail (no prefetch compares, 1024-thread blocks), and the overlap with a
                                 # tail that never waits inside the launch for a prefetch in flight (it leaves and comes back)
                                 {"DADA2HIP_V3_OVERLAP": "0", "DADA2HIP_V3_GRID": "2"}, {"DADA2HIP_V3_PF_WAIT_US": "0", "DADA2HIP_V3_GRID": "3"},
                                 # the round's evaluation riding on EVERY shuffle call behind the commit's (void attempts, locks taken back),
                                 # and never (a phase of its own: round 4's form); the default attempts behind calls that moved <= 16 uniques
                                 {"DADA2HIP_V3_SPEC_MAX": "1000000", "DADA2HIP_V3_GRID": "4"}, {"DADA2HIP_V3_SPEC": "0", "DADA2HIP_V3_GRID": "3"},
                                 # ... and only behind calls that moved <= 2 uniques: standing attempts, void attempts and plain calls in one round
                                 {"DADA2HIP_V3_SPEC_MAX": "2", "DADA2HIP_V3_GRID": "3"},
                                 # the batch screen without its presence bitmaps and the batch aligner without its pointer-free first pass
                                 {"DADA2HIP_SCREEN_BITS": "0", "DADA2HIP_AD_FAST": "0", "DADA2HIP_V3_GRID": "2"},
                                 # the XCD-hierarchical grid barrier (the default from 48 blocks on) on seven blocks in uneven groups: the
                                 # emulated XCC ids are not in block order and one XCC stays empty
                                 {"DADA2HIP_V3_XBAR": "1", "DADA2HIP_V3_GRID": "7"}, {"DADA2HIP_V3_XBAR": "1", "DADA2HIP_V3_GRID": "5", "DADA2HIP_V2_MOV_INLINE": "8", "DADA2HIP_V3_RING": "2"},
                                 # the launch chains (DADA2HIP_V2_TAIL=chain: what a second sample on the same device runs on)
                                 {"DADA2HIP_V2_TAIL": "chain"}, {"DADA2HIP_V2_TAIL": "chain", "DADA2HIP_V2_ALIGN": "commit"},
                                 {"DADA2HIP_V2_TAIL": "chain", "DADA2HIP_V2_LITE": "0", "DADA2HIP_V2_NBUF": "1", "DADA2HIP_V2_MOV_INLINE": "8"},
                                 {"DADA2HIP_V2_TAIL": "chain", "DADA2HIP_V2_ALIGN": "commit", "DADA2HIP_V2_NBUF": "1", "DADA2HIP_V2_CHAIN": "1",
                                  "DADA2HIP_NODE_CAP": "1"}],
                         ids=["default", "classic-engine", "lane-kernel", "wide-kernel", "align-at-commit-grid2", "tail-grid3", "tail-grid5-pauses",
                              "tail-grid2-ring1-nbuf1-grow", "tail-serial-grid2", "tail-overlap-no-wait-grid3", "tail-evaluate-on-every-call-grid4",
                              "tail-evaluate-apart-grid3", "tail-attempts-and-plain-calls-mixed-grid3", "exact-screen-full-aligner-grid2", "tail-xcd-barrier-grid7", "tail-xcd-barrier-grid5-pauses-ring2", "chains", "chains-align-at-commit", "chains-nolite-nbuf1-biglists",
                              "chains-commit-nbuf1-chain1-grow"])
def test_emulated_kernels_reproduce_the_reference_goldens(emu_lib, env):
    out = run_cases(emu_lib, ("sam1F_default", "sam1R_default") if not env else ("sam1F_default",), env)   # (CPU suite budget: both only once)
    assert "ok sam1F_default 10 " in out
    assert ("xcd-barrier" in out) == (env.get("DADA2HIP_V3_XBAR") == "1")   # (the small grids of these cases take the flat barrier unless told otherwise)


def test_emulated_long_read_band32_case(emu_lib):
    run_cases(emu_lib, ("samPB_band32",))


def test_emulated_bimera_pair_quantities_both_kernels(emu_lib):
    """The bimera mode of k_nw_ad (alignment reduced to get_lr / get_ham_endsfree inside the kernel, from column bitmaps) and the
    lane kernel + k_bimera_lr behind DADA2HIP_NW_KERNEL=lane, per alignment against the oracle."""
    code = (
        "import os, sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from dada2_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "import numpy as np\n"
        "from dada2_amd import api\n"
        "from oracle import cport\n"
        "from helpers import BIMERA_PAIR_OPTIONS, bimera_pair_cases\n"
        "for seed, n, L in ((1, 90, 60), (2, 60, 130), (3, 45, 251)):\n"
        "    qs, ps = bimera_pair_cases(seed, n, L)\n"
        "    for oo, ms, sc in BIMERA_PAIR_OPTIONS:\n"
        "        want = cport.bimera_pairs(qs, ps, oo, *sc, ms)\n"
        "        for kern in ('', 'lane'):\n"
        "            os.environ['DADA2HIP_NW_KERNEL'] = kern\n"
        "            got = api.bimera_pairs(qs, ps, oo, *sc, ms)\n"
        "            assert np.array_equal(got, want), (seed, oo, ms, sc, kern, np.nonzero((got != want).any(axis=1))[0][:5])\n"
        "from helpers import bimera_short_pair_cases\n"
        "qs, ps = bimera_short_pair_cases(5, 150)\n"
        "for oo, ms in ((True, 16), (False, 4), (True, 1), (True, 40)):\n"
        "    for kern in ('', 'lane'):\n"
        "        os.environ['DADA2HIP_NW_KERNEL'] = kern\n"
        "        assert np.array_equal(api.bimera_pairs(qs, ps, oo, max_shift=ms), cport.bimera_pairs(qs, ps, oo, max_shift=ms)), (oo, ms, kern)\n"
        "print('bimera pairs: ok')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), emu_lib)
    out = _run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "bimera pairs: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_emulated_bimera_table_and_nwvec_goldens(emu_lib):
    out = _run([sys.executable, os.path.join(ROOT, "tools", "emu_bimera.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "bimera table goldens: ok" in out.stdout and "nwvec goldens: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("env", [{}, {"DADA2HIP_AD_HOMO": "0"}], ids=["anti-diagonal-kernel", "lane-kernel"])
def test_emulated_homopolymer_gap_goldens(emu_lib, env):
    """HOMOPOLYMER_GAP_PENALTY: since round 4 on k_nw_ad<.., HOMO> (engine v2 and the persistent tail with it); DADA2HIP_AD_HOMO=0
    = the lane kernels and the classic engine as before."""
    run_cases(emu_lib, ("sam1F_homogap",), env)


def test_emulated_homopolymer_rich_samples_on_the_anti_diagonal_kernel(emu_lib):
    """helpers.homopolymer_sample (run-length errors: the option changes the result) on the edge geometry and at band 32 with a
    free homopolymer gap, against the oracle."""
    code = (
        "import sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from dada2_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "from helpers import HOMO_OPTION_CASES, homopolymer_sample, assert_results_equal, tperr1\n"
        "from dada2_amd import api\n"
        "from dada2_amd.io import extend_err\n"
        "from dada2_amd.opts import DadaOpts\n"
        "from oracle import cport\n"
        "for seed, kw in (HOMO_OPTION_CASES[1], HOMO_OPTION_CASES[3]):\n"
        "    d = homopolymer_sample(seed, nreads=3000)\n"
        "    err = extend_err(tperr1(), 40)\n"
        "    got = api.dada_uniques(d.seqs, d.abundances, None, err, d.quals, DadaOpts(**kw))\n"
        "    assert_results_equal(got, cport.dada_uniques(d.seqs, d.abundances, None, err, d.quals, DadaOpts(**kw)))\n"
        "    assert got.stats['tail_launches'] > 0\n"
        "print('homopolymer samples: ok')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), emu_lib)
    out = _run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "homopolymer samples: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_emulated_exact_bud_ties_settled_on_the_device_or_the_host(emu_lib):
    """tests/helpers.zero_tie_sample: ties of b_bud's best key that k2_birth settles itself (first-slot members of partition 0,
    one candidate in the lowest partition) next to those it must leave to the host (moved members, several in one partition)."""
    out = _run([sys.executable, os.path.join(ROOT, "tools", "emu_ties.py"), "1", "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "zero ties: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_emulated_merge_pairs_goldens_with_packed_pairs(emu_lib):
    """dada2hip_merge_pairs on the emulated library: unbanded alignments on the lane kernel with a centre per work item (64
    unrelated pairs to a wave), rows equal to the goldens made with the reference's C_nwalign / C_eval_pair / C_pair_consensus."""
    out = _run([sys.executable, os.path.join(ROOT, "tools", "emu_merge.py"), "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "merge goldens: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def _seeded_through_emulator(emu_lib, cases, env=None, timeout=1500):
    code = (
        "import sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from dada2_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "from helpers import seeded_option_sample, assert_results_equal, tperr1\n"
        "from dada2_amd import api\n"
        "from dada2_amd.opts import DadaOpts\n"
        "from oracle import cport\n"
        "for seed, kw in %r:\n"
        "    d, pri = seeded_option_sample(seed)\n"
        "    o = DadaOpts(**kw)\n"
        "    got = api.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o)\n"
        "    want = cport.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o)\n"
        "    assert_results_equal(got, want, check_birth_from=pri is None)\n"
        "    print('ok', seed, got.nclust, got.stats['nnw'])\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), emu_lib, list(cases))
    e = dict(os.environ)
    e.update(env or {})
    out = _run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and out.stdout.count("ok ") == len(cases), out.stdout[-2000:] + out.stderr[-4000:]


# the band / geometry cases tools/emu_seeded.py used to hold (the only non-default-score runs lived in a dev tool)
EMU_GEOMETRY_CASES = [(10, dict(BAND_SIZE=0)), (11, dict(BAND_SIZE=-1)), (12, dict(BAND_SIZE=40)), (15, dict(BAND_SIZE=1)), (16, dict(BAND_SIZE=18)),
                      (17, dict(BAND_SIZE=19)), (18, dict(BAND_SIZE=20)),
                      (20, dict(HOMOPOLYMER_GAP_PENALTY=-1)), (21, dict(HOMOPOLYMER_GAP_PENALTY=-1, BAND_SIZE=32)),
                      (22, dict(HOMOPOLYMER_GAP_PENALTY=-2, BAND_SIZE=-1)), (23, dict(HOMOPOLYMER_GAP_PENALTY=0, GAP_PENALTY=-6))]


@pytest.mark.parametrize("nw_kernel", ["coop", "lane", "wide"])
def test_emulated_option_sweep_with_user_scores_and_sse1(emu_lib, nw_kernel):
    """tests/helpers.py's seeded option sweep through the emulated library on each aligner family - including MATCH / MISMATCH /
    GAP_PENALTY away from 5 / -4 / -8 (the general-score instances of k_nw_ad) and SSE = 1."""
    from helpers import BASE_OPTION_CASES, SCORE_OPTION_CASES
    cases = (BASE_OPTION_CASES if nw_kernel == "coop" else []) + SCORE_OPTION_CASES   # (the base sweep once; the -m gpu tests run it on every family)
    _seeded_through_emulator(emu_lib, cases, {"DADA2HIP_NW_KERNEL": nw_kernel})


def test_emulated_band_geometries(emu_lib):
    _seeded_through_emulator(emu_lib, EMU_GEOMETRY_CASES, {"DADA2HIP_NW_KERNEL": "coop"})


HOOKS_CODE = (
    "import sys\n"
    "sys.path[:0] = [%r, %r]\n"
    "from dada2_amd import _lib\n"
    "if %r: _lib.LIB_PATH = %r\n"
    "from helpers import case_inputs, assert_results_equal\n"
    "from dada2_amd import api\n"
    "d, err, pri, o, exp, meta = case_inputs('sam1F_default')\n"
    "for k in %r:\n"
    "    polls = []\n"
    "    def stop():\n"
    "        polls.append(1)\n"
    "        return len(polls) >= k\n"
    "    try:\n"
    "        api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o, should_abort=stop)\n"
    "        raise SystemExit('not aborted')\n"
    "    except _lib.Dada2HipError as ex:\n"
    "        assert ex.code == 5 and 'aborted' in str(ex), (ex.code, str(ex))\n"
    "    assert len(polls) == k, (len(polls), k)\n"
    "    got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)\n"          # the same process, right behind the abort
    "    assert_results_equal(got, exp)\n"
    "lines = []\n"
    "got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o, verbose=True, log=lines.append, should_abort=lambda: False)\n"
    "assert_results_equal(got, exp)\n"
    "text = ''.join(lines)\n"
    "assert text.count('New Cluster C') == got.nclust - 1, text\n"                      # Rmain.cpp:317, once per birth
    "assert 'ALIGN: 8655 aligns, 3032 shrouded (%%d raw).' %% len(d.seqs) in text, text\n"   # Rmain.cpp:333 with the reference's counters
    "class Boom(Exception): pass\n"
    "def bad(): raise Boom('from the callback')\n"
    "try:\n"
    "    api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o, should_abort=bad)\n"
    "    raise SystemExit('callback exception lost')\n"
    "except Boom: pass\n"
    "assert_results_equal(api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o), exp)\n"
    "print('hooks: ok')\n"
)

# (the CPU suite's budget: the persistent tail with several blocks, the chains and the classic engine; the -m gpu test adds the
#  one-block default)
ENGINE_ENVS = [{"DADA2HIP_V3_GRID": "3"}, {"DADA2HIP_V2_TAIL": "chain"}, {"DADA2HIP_ENGINE": "classic"}]
ENGINE_IDS = ["persistent-tail-grid3", "chains", "classic-engine"]
ENGINE_ABORT_POLLS = [(1, 4), (4,), (1,)]   # (the CPU suite's budget: both polls on the persistent tail, one each on the others)


@pytest.mark.parametrize("env,polls", list(zip(ENGINE_ENVS, ENGINE_ABORT_POLLS)), ids=ENGINE_IDS)
def test_emulated_abort_hook_and_verbose_log_on_every_engine(emu_lib, env, polls):
    """dada2hip_hooks (Rcpp::checkUserInterrupt / the verbose Rprintfs, src/Rmain.cpp:317-333): a run aborted at its first and
    fourth round returns DADA2HIP_ERR_ABORTED with launches still queued, the next run in the same process equals the golden; the
    log carries one line per birth and the reference's nalign / nshroud."""
    code = HOOKS_CODE % (ROOT, os.path.join(ROOT, "tests"), True, emu_lib, polls)
    e = dict(os.environ)
    e.update(env)
    out = _run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "hooks: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("env", [{"DADA2HIP_V3_FAIL_ENTRY": "1"}, {"DADA2HIP_V3_FAIL_ENTRY": "2", "DADA2HIP_V3_GRID": "3"},
                                 {"DADA2HIP_V3_FAIL_ENTRY": "3", "DADA2HIP_V2_NBUF": "1"}],
                         ids=["first-launch", "second-launch-grid3", "third-launch-nbuf1"])
def test_emulated_entry_barrier_failure_continues_on_the_launch_chains(emu_lib, env):
    """A persistent launch whose blocks do not all become resident gives up at its entry barrier, before anything has changed
    (DADA2HIP_V3_FAIL_ENTRY=n makes the n-th launch do exactly that): the run goes on on the launch chains and equals the golden."""
    code = (
        "import sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from dada2_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "from helpers import case_inputs, assert_results_equal\n"
        "from dada2_amd import api\n"
        "d, err, pri, o, exp, meta = case_inputs('sam1F_default')\n"
        "got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)\n"
        "assert_results_equal(got, exp)\n"
        "assert got.stats['tail_fallbacks'] == 1, got.stats['tail_fallbacks']\n"
        "print('fallback: ok', got.stats['tail_launches'])\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), emu_lib)
    e = dict(os.environ)
    e.update(env)
    out = _run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "fallback: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("env", [{"DADA2HIP_V3_GRID": "4", "DADA2HIP_V2_NBUF": "4", "DADA2HIP_V3_RING": "2"},
                                 {"DADA2HIP_V3_GRID": "2", "DADA2HIP_V2_NBUF": "5", "DADA2HIP_V2_MOV_INLINE": "16", "DADA2HIP_V3_PF_WAIT_US": "0"},
                                 # round 5's form of the tail under the overlap: 512-thread blocks beside the compares on every CU
                                 {"DADA2HIP_V3_GRID": "3", "DADA2HIP_V3_BLOCK": "512"},
                                 # round 6: the tail's LDS mirror compared with the state arrays at the end of every round
                                 # (DADA2HIP_V3_MIRROR=2), with pauses and attempts / plain calls mixed; and the tail without it
                                 {"DADA2HIP_V3_GRID": "3", "DADA2HIP_V3_MIRROR": "2", "DADA2HIP_V2_MOV_INLINE": "16", "DADA2HIP_V3_SPEC_MAX": "3"},
                                 {"DADA2HIP_V3_GRID": "2", "DADA2HIP_V3_MIRROR": "0"},
                                 # the replay lane (a second host thread replays the published moves and births), with pauses
                                 {"DADA2HIP_V3_GRID": "3", "DADA2HIP_V3_LANE": "1", "DADA2HIP_V2_MOV_INLINE": "16"},
                                 # every mover list through the radix sort of the host's replay (default: lists of 4 096 movers and more)
                                 {"DADA2HIP_V3_GRID": "2", "DADA2HIP_REPLAY_RADIX_MIN": "1"}],
                         ids=["grid4-nbuf4-ring2", "grid2-nbuf5-pauses-no-wait", "grid3-512-thread-blocks", "grid3-mirror-checked-pauses", "grid2-no-mirror",
                              "grid3-replay-lane-pauses", "grid2-replay-radix-always"])
def test_emulated_prefetch_compares_under_the_tail_on_a_deeper_sample(emu_lib, env):
    """The next batch's compare planned by the persistent tail and run on the second stream on the 20-partition
    golden: rounds served out of prefetched batches, with the smallest cache that allows it (the buffer whose rows the coming
    round commits is never the one a prefetch recycles), with pauses, with a host that lags."""
    code = (
        "import sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from dada2_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "from helpers import case_inputs, assert_results_equal\n"
        "from dada2_amd import api\n"
        "d, err, pri, o, exp, meta = case_inputs('synth3000_default')\n"
        "got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)\n"
        "assert_results_equal(got, exp)\n"
        "st = got.stats\n"
        "assert st['overlap_on'] == 1 and st['pf_compares'] >= 1 and st['pf_hits'] >= 2, {k: st[k] for k in ('overlap_on', 'pf_compares', 'pf_hits')}\n"
        "assert st['tail_threads'] == %d and st['tail_mirror'] == %d, (st['tail_threads'], st['tail_mirror'])\n"
        "print('prefetch: ok', st['pf_compares'], st['pf_hits'], st['pf_exits'])\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), emu_lib, int(env.get("DADA2HIP_V3_BLOCK", "1024")), 0 if env.get("DADA2HIP_V3_MIRROR") == "0" else 1)
    e = dict(os.environ)
    e.update(env)
    out = _run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "prefetch: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_emulated_run_without_the_persistent_slot_takes_the_chains_and_plans_no_prefetch(emu_lib, tmp_path):
    """Another process holds the device's persistent slot (its lock file): the run takes the launch chains - and must not carry
    the persistent tail's prefetch planner with it (a round would take its comparisons from a batch nobody ever compared: found
    on the MI355X by three ranks sharing one GPU)."""
    import fcntl
    pci = "emutest_%d" % os.getpid()
    lock = open("/tmp/dada2hip_persistent_%s.lock" % pci, "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        code = (
            "import sys\n"
            "sys.path[:0] = [%r, %r]\n"
            "from dada2_amd import _lib\n"
            "_lib.LIB_PATH = %r\n"
            "from helpers import case_inputs, assert_results_equal\n"
            "from dada2_amd import api\n"
            "d, err, pri, o, exp, meta = case_inputs('synth3000_default')\n"
            "got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, o)\n"
            "assert_results_equal(got, exp)\n"
            "st = got.stats\n"
            "assert st['tail_launches'] == 0 and st['overlap_on'] == 0 and st['pf_compares'] == 0, st\n"
            "print('chains without the slot: ok')\n"
        ) % (ROOT, os.path.join(ROOT, "tests"), emu_lib)
        e = dict(os.environ)
        e["EMU_PCI_ID"] = pci
        out = _run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "chains without the slot: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()
        os.remove("/tmp/dada2hip_persistent_%s.lock" % pci)


NWVEC_LETTERS_CODE = (
    "import sys\n"
    "sys.path[:0] = [%r, %r]\n"
    "from dada2_amd import _lib\n"
    "if %r: _lib.LIB_PATH = %r\n"
    "from helpers import nwvec_letter_cases\n"
    "from dada2_amd import api\n"
    "from oracle import ref\n"
    "s1, s2 = nwvec_letter_cases()\n"
    "for band, ef, sc in ((16, True, (5, -4, -8)), (-1, True, (5, -4, -8)), (8, False, (5, -4, -8)), (16, True, (1, -1, -2))):\n"
    "    got = api.nwvec(s1, s2, sc[0], sc[1], sc[2], band, ef)\n"
    "    for i, (a, b) in enumerate(zip(s1, s2)):\n"
    "        assert tuple(got[i]) == ref.nwvec_raw(a, b, sc[0], sc[1], sc[2], band, ef), (band, ef, sc, a, b, got[i])\n"
    "try:\n"
    "    api.nwvec(['ABCDEFGHIJKLMNOPQRS'], ['ABCDEFGHIJKLMNOPQ'])\n"
    "    raise SystemExit('17 letters accepted')\n"
    "except _lib.Dada2HipError as ex:\n"
    "    assert ex.code == 4, ex.code\n"
    "try:\n"
    "    api.nwalign('ACGTN', 'ACGT')\n"                       # C_nwalign: the reference's behaviour for N is undefined - still refused
    "    raise SystemExit('nwalign took N')\n"
    "except _lib.Dada2HipError as ex:\n"
    "    assert ex.code == 4, ex.code\n"
    "print('nwvec letters: ok', len(s1))\n"
)


def test_emulated_nwvec_on_letters_outside_acgt_matches_the_reference(emu_lib, oracle_ref):
    """C_nwvec compares the strings' raw bytes (nwalign_vectorized.cpp:165): N, IUPAC codes, lower case, anything.  The device
    path renumbers each pair's letters (<= 16) into two 2-bit planes; every pair against the reference's own call on raw bytes."""
    code = NWVEC_LETTERS_CODE % (ROOT, os.path.join(ROOT, "tests"), True, emu_lib)
    out = _run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "nwvec letters: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_emulated_quality_marshalling_vector_sweep_and_its_scalar_redo(emu_lib):
    """raw_new's (uint8) round(mean quality) (containers.cpp:34) at the boundary: the branch-free sweep (hostsimd.cpp) for rows in
    [0, 255.5), the exact scalar rule for rows that hold anything else - a value that rounds to -0 stays valid, negative / too
    large / NaN inside a read are refused with the reference's message."""
    code = (
        "import sys\n"
        "sys.path[:0] = [%r, %r]\n"
        "from dada2_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "import numpy as np\n"
        "from helpers import case_inputs, assert_results_equal\n"
        "from dada2_amd import api\n"
        "from oracle import cport\n"
        "d, err, pri, o, exp, meta = case_inputs('sam1F_default')\n"
        "for val in (-1.0, 255.6, float('nan')):\n"
        "    q = d.quals.copy(); q[5, 10] = val\n"
        "    try:\n"
        "        api.dada_uniques(d.seqs, d.abundances, pri, err, q, o)\n"
        "        raise SystemExit('accepted %%r' %% val)\n"
        "    except _lib.Dada2HipError as ex:\n"
        "        assert ex.code == 1 and 'Invalid derep$quals matrix' in str(ex), str(ex)\n"
        "q = d.quals.copy(); q[7, 3] = -0.2; q[9, 0] = 30.5; q[11, 1] = 29.4999\n"
        "got = api.dada_uniques(d.seqs, d.abundances, pri, err, q, o)\n"
        "assert_results_equal(got, cport.dada_uniques(d.seqs, d.abundances, pri, err, q, o))\n"
        "print('quality marshalling: ok')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), emu_lib)
    out = _run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "quality marshalling: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
