#!/usr/bin/env python3
"""tools/emu_bimera.py - development aid: the bimera goldens (tests/golden/bimera_table.npz, produced by the reference's
chimera.cpp) and the mergePairs goldens through the EMULATED library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
import build as emu_build  # noqa: E402
from dada2_amd import _lib  # noqa: E402

_lib.LIB_PATH = emu_build.build()
import numpy as np  # noqa: E402
from dada2_amd import api  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "bimera_table.npz"))
mat, seqs = z["mat"], [str(s) for s in z["seqs"]]
for oo in (0, 1):
    for ms in (16, 4):
        nflag, nsam = api.table_bimera2(mat, seqs, allow_one_off=bool(oo), max_shift=ms)
        assert np.array_equal(nflag, z[f"nflag_oo{oo}_ms{ms}"]), (oo, ms)
        assert np.array_equal(nsam, z[f"nsam_oo{oo}_ms{ms}"])
print("bimera table goldens: ok")
z = np.load(os.path.join(ROOT, "tests", "golden", "nwalign_pairs.npz"))
s1, s2, band = [str(x) for x in z["s1"]], [str(x) for x in z["s2"]], z["band"]
for b in sorted(set(band.tolist())):
    idx = np.nonzero(band == b)[0][:40]
    got = api.nwvec([s1[i] for i in idx], [s2[i] for i in idx], 5, -4, -8, int(b))
    for k, i in enumerate(idx):
        assert got[k] == (str(z["al0"][i]), str(z["al1"][i])), (i, b)
print("nwvec goldens: ok")
