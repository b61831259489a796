#!/usr/bin/env python3
"""tools/emu_check.py - development aid: runs engine-heavy seeded samples through the EMULATED library (tests/emu) and
compares every output with the plain-C oracle.  Usage: python tools/emu_check.py [n:G:L ...]   (default 6000:48:200 20000:96:250)
Environment knobs of the library (DADA2HIP_*) apply; DADA2HIP_NW_KERNEL=lane makes the emulation fast (no cross-lane traffic)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
import build as emu_build  # noqa: E402
from dada2_amd import _lib  # noqa: E402

_lib.LIB_PATH = emu_build.build()
from helpers import assert_results_equal, tperr1  # noqa: E402
from dada2_amd import api  # noqa: E402
from dada2_amd.opts import DadaOpts  # noqa: E402
from dada2_amd.synth import make_sample  # noqa: E402
from oracle import cport  # noqa: E402

specs = sys.argv[1:] or ["6000:48:200", "20000:96:250"]
for spec in specs:
    n, G, L = (int(x) for x in spec.split(":"))
    d = make_sample(tperr1(), n, L=L, G=G, seed=7000 + n % 997, chunk=max(4000, n))
    t0 = time.time()
    got = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts())
    t1 = time.time()
    want = cport.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts())
    assert_results_equal(got, want)
    st = got.stats
    print(f"{spec}: ok, {got.nclust} partitions, emulated in {t1 - t0:.1f} s (oracle {time.time() - t1:.1f} s); nnw {st['nnw']} shuffles {st['nshuffle']} "
          f"moves {st['nmoves']} batch compares {st['batch_compares']}; aligner ran {st['nnw_run']} + {st['ngapless_run']} gapless for the rounds", flush=True)
