#!/usr/bin/env python3
"""tools/emu_ties.py - development aid / CPU test leg: samples made of exact b_bud ties (tests/helpers.zero_tie_sample)
through the EMULATED library (tests/emu), every output against the plain-C oracle.  DADA2HIP_V2_SUMMARY=1 prints which
ties the device settled and which went to the host."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
import build as emu_build  # noqa: E402
from dada2_amd import _lib  # noqa: E402

_lib.LIB_PATH = emu_build.build()
from helpers import assert_results_equal, tperr1, zero_tie_sample  # noqa: E402
from dada2_amd import api  # noqa: E402
from dada2_amd.opts import DadaOpts  # noqa: E402
from oracle import cport  # noqa: E402

seeds = [int(x) for x in sys.argv[1:]] or [1, 2, 3]
for seed in seeds:
    seqs, ab, q = zero_tie_sample(seed)
    for o in (DadaOpts(), DadaOpts(OMEGA_A=1e-4, DETECT_SINGLETONS=True)):
        got = api.dada_uniques(seqs, ab, None, tperr1(), q, o)
        want = cport.dada_uniques(seqs, ab, None, tperr1(), q, o)
        assert_results_equal(got, want)
        print("seed", seed, "ok:", got.nclust, "partitions of", len(seqs), "uniques", flush=True)
print("zero ties: ok")
