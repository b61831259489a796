#!/usr/bin/env python3
"""tools/emu_merge.py - development aid / CPU test leg: the mergePairs goldens (tests/golden/merge_*.npz, made with the
reference's own C_nwalign / C_eval_pair / C_pair_consensus) through the EMULATED library: dada2hip_merge_pairs with its
unbanded alignments on the lane kernel, 64 unrelated pairs to a wave (NwArgs::pair_centre)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
import build as emu_build  # noqa: E402
from dada2_amd import _lib  # noqa: E402

_lib.LIB_PATH = emu_build.build()
from test_merge import OPTION_SETS, _call_merge, assert_rows_equal, golden_rows, make_case  # noqa: E402

seeds = [int(x) for x in sys.argv[1:]] or [1, 2]
for seed in seeds:
    c = make_case(seed)
    for name in sorted(OPTION_SETS):
        kw = dict(min_overlap=12, max_mismatch=0, trim_overhang=False, just_concatenate=False)
        kw.update(OPTION_SETS[name])
        rc, msg, rows = _call_merge(c, **kw)
        assert rc == 0, msg
        assert_rows_equal(rows, golden_rows(seed, name))
        print("seed", seed, name, "ok:", len(rows), "rows", flush=True)
print("merge goldens: ok")
