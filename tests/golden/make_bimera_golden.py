#!/usr/bin/env python3
"""Generate tests/golden/bimera_table.npz from the REFERENCE ITSELF (src/chimera.cpp compiled in place into oracle/_ref).
Run in the authoring container only:   make -C oracle && python tests/golden/make_bimera_golden.py

Input: the sequence table of the reference's own paired fixtures after denoising (ASVs of sam1F / sam2F / sam1R / sam2R as the
committed goldens give them, one table per read direction stacked as 4 "samples" over the union of sequences is not
meaningful, so: 2 samples x F ASVs), extended with constructed bimeras of the most abundant ASVs - exact, one-off, shifted,
with an internal indel - at low abundance, which is what removeBimeraDenovo exists to find.  Output: nflag / nsam of
C_table_bimera2 for both allowOneOff settings and two maxShift values, plus C_is_bimera of every sequence against the more
abundant ones."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.abspath(os.path.join(HERE, "..", "..")), os.path.abspath(os.path.join(HERE, ".."))]
from helpers import case_inputs  # noqa: E402
from oracle import ref  # noqa: E402


def build_table():
    rng = np.random.default_rng(20260925)
    tabs = []
    for name in ("sam1F_default", "sam2F_nogreedy"):
        d, err, pri, opts, exp, meta = case_inputs(name)
        tabs.append(dict(zip(exp.clustering["sequence"], exp.clustering["abundance"].tolist())))
    seqs = sorted(set(tabs[0]) | set(tabs[1]), key=lambda s: (-(tabs[0].get(s, 0) + tabs[1].get(s, 0)), s))
    par = seqs[:8]
    extra = []
    for t in range(40):
        a, b = rng.choice(len(par), 2, replace=False)
        cut = int(rng.integers(30, 220))
        ch = par[a][:cut] + par[b][cut:]
        kind = t % 5
        if kind == 1:      # one-off
            p = int(rng.integers(5, len(ch) - 5))
            ch = ch[:p] + "ACGT"[("ACGT".index(ch[p]) + 1 + int(rng.integers(0, 3))) % 4] + ch[p + 1:]
        elif kind == 2:    # shifted start
            ch = ch[int(rng.integers(1, 12)):]
        elif kind == 3:    # internal deletion next to the junction
            ch = ch[:cut + 7] + ch[cut + 8:]
        elif kind == 4:    # three-parent mosaic: not a bimera
            c = int(rng.choice(len(par)))
            cut2 = min(len(ch) - 20, cut + int(rng.integers(30, 80)))
            ch = ch[:cut2] + par[c][cut2:]
        if ch not in seqs and ch not in extra:
            extra.append(ch)
    allseq = seqs + extra
    mat = np.zeros((2, len(allseq)), dtype=np.int32)
    for i, t in enumerate(tabs):
        for j, s in enumerate(allseq):
            mat[i, j] = t.get(s, 0)
    for j in range(len(seqs), len(allseq)):
        mat[:, j] = rng.integers(0, 9, size=2)
        if mat[:, j].sum() == 0:
            mat[0, j] = 3
    return mat, allseq


def main():
    mat, seqs = build_table()
    out = {"mat": mat, "seqs": np.array(seqs)}
    for oo in (0, 1):
        for ms in (16, 4):
            nflag, nsam = ref.table_bimera2(mat, seqs, min_fold=1.5, min_abund=2, allow_one_off=bool(oo), min_one_off_par_dist=4,
                                            max_shift=ms)
            out[f"nflag_oo{oo}_ms{ms}"] = nflag
            out[f"nsam_oo{oo}_ms{ms}"] = nsam
        tot = mat.sum(axis=0)
        isb = []
        for j, s in enumerate(seqs):
            pars = [seqs[k] for k in range(len(seqs)) if tot[k] > 2 * tot[j] and tot[k] > 8]
            isb.append(ref.is_bimera(s, pars, allow_one_off=bool(oo)))
        out[f"isbim_oo{oo}"] = np.array(isb)
    np.savez_compressed(os.path.join(HERE, "bimera_table.npz"), **out)
    print(len(seqs), "sequences;", {k: v.tolist() for k, v in out.items() if k.startswith("nflag")})


if __name__ == "__main__":
    main()
