"""tests/at_size.py — TEST INFRASTRUCTURE: the reference's results for BASELINE.json's configurations at their stated sizes,
computed in BACKGROUND processes while the rest of the `-m gpu` suite runs (the three at-size tests spent 340 of
the suite's 544 s drawing their samples and waiting for the reference on all cores).

`start()` (called once per session by conftest.py when GPU tests are selected and oracle/_ref is there) launches one worker per
case; a worker draws the case's synthetic sample(s) exactly as bench.py does (same seeds, through bench.py's input cache), runs
the reference binary itself (oracle/_ref, multithread=TRUE on a share of the host's cores) and pickles its results.  `get(case)`
waits for the worker and returns (dereps, err, opts, reference results); the test then runs the GPU on the same inputs and
compares every output.  Nothing here is reachable from dada2_amd/."""
import os
import pickle
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.environ.get("DADA2HIP_TEST_CACHE", "/tmp/dada2hip_test_cache")
CASES = ("cfg3", "cfg4", "cfg5")
_procs = {}


def _opts(case):
    from dada2_amd.opts import DadaOpts
    return {"cfg3": DadaOpts(), "cfg4": DadaOpts(), "cfg5": DadaOpts(BAND_SIZE=32, MAX_CLUST=32)}[case]


def _inputs(case):
    from types import SimpleNamespace
    sys.path.insert(0, ROOT)
    import bench
    a = SimpleNamespace(uniques=int(os.environ.get("DADA2HIP_TEST_UNIQUES", "0")), length=0, variants=0, deep=False)   # (override: dry runs of this file)
    dereps, _, err, _, _ = bench.make_inputs(int(case[3:]), a, 0, host_inputs=False)
    return dereps, err


def _path(case):
    return os.path.join(CACHE, f"ref_{case}.pkl")


def start(cases=CASES):
    """Launch the workers (idempotent).  Each gets a share of the host's cores: the reference's own thread sweep peaks at 32-64
    threads on these samples (bench.py cpu_baseline.sweep), more only adds scheduling noise."""
    os.makedirs(CACHE, exist_ok=True)
    share = max(8, min(64, (os.cpu_count() or 8) // max(1, len(cases))))
    for c in cases:
        if c in _procs:
            continue
        try:
            os.remove(_path(c))
        except OSError:
            pass
        env = dict(os.environ)
        env["HIP_VISIBLE_DEVICES"] = ""          # (a worker never touches the GPU)
        env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
        log = open(os.path.join(CACHE, f"ref_{c}.log"), "w")
        _procs[c] = subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", c, str(share)], env=env, stdout=log, stderr=log)


def get(case, timeout=1500):
    """(dereps, err, opts, [reference result per sample]) of `case`; runs the worker now if nobody started it."""
    if case not in _procs:
        start((case,))
    p = _procs[case]
    t0 = time.time()
    while p.poll() is None:
        if time.time() - t0 > timeout:
            p.kill()
            raise RuntimeError(f"reference worker for {case} timed out")
        time.sleep(0.2)
    if p.returncode != 0:
        raise RuntimeError(f"reference worker for {case} failed: see {os.path.join(CACHE, 'ref_' + case + '.log')}\n" +
                           open(os.path.join(CACHE, f"ref_{case}.log")).read()[-2000:])
    with open(_path(case), "rb") as fh:
        want = pickle.load(fh)
    dereps, err = _inputs(case)                   # (the worker drew them into bench.py's input cache)
    return dereps, err, _opts(case), want


def _worker(case, threads):
    from oracle import ref
    assert ref.available(), "oracle/_ref not built"
    dereps, err = _inputs(case)
    o = _opts(case)
    ref.set_threads(threads)
    t0 = time.time()
    out = [ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o, multithread=True) for d in dereps]
    print(f"{case}: {len(dereps)} sample(s), reference took {time.time() - t0:.1f} s on {threads} threads", flush=True)
    tmp = _path(case) + f".{os.getpid()}.tmp"
    with open(tmp, "wb") as fh:
        pickle.dump(out, fh, protocol=pickle.HIGHEST_PROTOCOL)
    os.replace(tmp, _path(case))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "worker":
        _worker(sys.argv[2], int(sys.argv[3]))
