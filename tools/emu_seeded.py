#!/usr/bin/env python3
"""tools/emu_seeded.py - development aid: the seeded option sweep of tests/test_gpu_parity.py::test_seeded_samples_match_oracle
(ragged lengths, indels, bands 0 / 4 / 16 / 40 / unbanded, non-default options) through the EMULATED library, every case on
the anti-diagonal kernel where it applies (DADA2HIP_NW_KERNEL=coop), against the plain-C oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
import build as emu_build  # noqa: E402
from dada2_amd import _lib  # noqa: E402

_lib.LIB_PATH = emu_build.build()
import numpy as np  # noqa: E402
from helpers import assert_results_equal, tperr1  # noqa: E402
from dada2_amd import api  # noqa: E402
from dada2_amd.opts import DadaOpts  # noqa: E402
from dada2_amd.synth import make_sample  # noqa: E402
from oracle import cport  # noqa: E402

from helpers import BAND_OPTION_CASES, BASE_OPTION_CASES, SCORE_OPTION_CASES  # noqa: E402
from test_emu import EMU_GEOMETRY_CASES  # noqa: E402

CASES = BASE_OPTION_CASES + BAND_OPTION_CASES + SCORE_OPTION_CASES + [c for c in EMU_GEOMETRY_CASES if c[0] > 12]   # (the tests run the same lists)
os.environ.setdefault("DADA2HIP_NW_KERNEL", "coop")
only = [int(x) for x in sys.argv[1:]]
for seed, kw in CASES:
    if only and seed not in only:
        continue
    ragged = seed % 2 == 0
    d = make_sample(tperr1(), 800, L=120, G=8, seed=seed, Lmin=100 if ragged else None, indel_rate=2e-3 if ragged else 0.0, chunk=4000)
    o = DadaOpts(**kw)
    pri = (np.arange(d.nraw) % 17 == 3).astype(np.uint8) if seed == 6 else None
    t0 = time.time()
    got = api.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o)
    want = cport.dada_uniques(d.seqs, d.abundances, pri, tperr1(), d.quals, o)
    assert_results_equal(got, want, check_birth_from=pri is None)
    print(f"seed {seed} {kw}: ok ({got.nclust} partitions, nnw {got.stats['nnw']}, {time.time() - t0:.1f} s)", flush=True)
