#!/usr/bin/env python3
"""Goldens for the two aligners beside nwalign_endsfree that C_nwalign / raw_align can reach (test infrastructure; needs
/root/reference and the built oracle/_ref):
  * nwalign_endsfree_homo (src/nwalign_endsfree.cpp:220-396; dada(HOMOPOLYMER_GAP_PENALTY=...), R/dada.R:222-231)
  * global nwalign        (src/nwalign_endsfree.cpp:403-537; C_nwalign(endsfree=FALSE), src/evaluate.cpp:44-48)
Everything stored here is the output of the reference's own C++ (oracle/_ref):
  <fixture>_homogap.expected.npz   whole dada_uniques results (read by tests/helpers.case_inputs)
  nwalign_variants.npz             random homopolymer-rich pairs with the alignments of C_nwalign for every variant"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
from make_golden import save_result  # noqa: E402
from helpers import load_input, tperr1  # noqa: E402
from dada2_amd.io import extend_err  # noqa: E402
from dada2_amd.opts import DadaOpts  # noqa: E402
from oracle import ref  # noqa: E402


def main():
    for name, fq, kw in (("sam1F_homogap", "sam1F", dict(HOMOPOLYMER_GAP_PENALTY=-1)),
                         ("samPB_homogap_band32", "samPB", dict(HOMOPOLYMER_GAP_PENALTY=-1, BAND_SIZE=32))):
        d = load_input(fq)
        err = extend_err(tperr1(), int(np.ceil(np.nanmax(d.quals))))
        o = DadaOpts(**kw)
        r = ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
        save_result(name, r, {"input": fq, "err": "tperr1", "opts": kw, "nclust": int(r.nclust)})
        print(name, r.nclust, r.clustering["abundance"].tolist()[:12])
    rng = np.random.default_rng(20260925)

    def rnd(L):
        s = []
        while len(s) < L:
            b = "ACGT"[int(rng.integers(4))]
            s += [b] * (int(rng.integers(3, 8)) if rng.random() < 0.3 else 1)
        return "".join(s[:L])

    def mutate(s):
        s = list(s)
        for _ in range(int(rng.integers(0, 7))):
            p = int(rng.integers(len(s)))
            k = int(rng.integers(3))
            if k == 0:
                s[p] = "ACGT"[int(rng.integers(4))]
            elif k == 1:
                del s[p]
            else:
                s.insert(p, s[p])
        return "".join(s)
    rows = []
    for t in range(300):
        a = rnd(int(rng.integers(12, 300)))
        b = mutate(a) if t % 7 else rnd(int(rng.integers(12, 300)))
        band = int(rng.choice([-1, 1, 4, 16, 32, 100]))
        hg = int(rng.choice([-8, -1, -2, 0, -4]))
        ef = bool(t % 3)
        al = ref.C_nwalign(a, b, 5, -4, -8, hg, band, ef)
        rows.append((a, b, band, hg, int(ef), al[0], al[1]))
    np.savez_compressed(os.path.join(HERE, "nwalign_variants.npz"), s1=np.array([r[0] for r in rows]), s2=np.array([r[1] for r in rows]),
                        band=np.array([r[2] for r in rows]), homo_gap=np.array([r[3] for r in rows]),
                        endsfree=np.array([r[4] for r in rows]), al0=np.array([r[5] for r in rows]), al1=np.array([r[6] for r in rows]))
    print("nwalign_variants:", len(rows), "pairs")


if __name__ == "__main__":
    main()
