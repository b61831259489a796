"""Synthetic mergePairs inputs shared by the golden generator and the tests: denoised forward / reverse sequence sets of
one sample cut from a few amplicons, and the per-read-pair indices into them (1-based, <= 0 for NA)."""
import numpy as np

_COMP = str.maketrans("ACGT", "TGCA")

OPTION_SETS = {
    "default": dict(),
    "mismatch1": dict(max_mismatch=1),
    "overlap40_trim": dict(min_overlap=40, trim_overhang=True),
    "concat": dict(just_concatenate=True),
}


def rc(s):
    return s.translate(_COMP)[::-1]


def make_case(seed, namp=14, npairs=900):
    rng = np.random.default_rng(seed)
    seqsF, seqsR = [], []
    truth = []                                             # (forward index, reverse index) that belong together
    for a in range(namp):
        L = int(rng.integers(150, 330))                    # amplicon length: overlaps from long to none, and overhangs
        amp = "".join(rng.choice(list("ACGT"), size=L))
        lf, lr = int(rng.integers(110, 160)), int(rng.integers(110, 160))
        f = amp[:lf] if lf <= L else amp + "".join(rng.choice(list("ACGT"), size=lf - L))    # read-through past the amplicon
        r_fw = amp[max(0, L - lr):]
        if lr > L:
            r_fw = "".join(rng.choice(list("ACGT"), size=lr - L)) + amp
        if a % 5 == 1:                                     # a substitution inside the reverse read: mismatch in the overlap
            p = 8 if lf + lr - L >= 30 else int(rng.integers(5, len(r_fw) - 5))      # (inside the overlap when there is one)
            r_fw = r_fw[:p] + "ACGT"[("ACGT".index(r_fw[p]) + 1) % 4] + r_fw[p + 1:]
        if a % 7 == 2:                                     # a deletion: indel in the overlap
            p = 14 if lf + lr - L >= 30 else int(rng.integers(10, len(r_fw) - 10))
            r_fw = r_fw[:p] + r_fw[p + 1:]
        seqsF.append(f)
        seqsR.append(rc(r_fw))
        truth.append((a + 1, a + 1))
    permF, permR = rng.permutation(namp), rng.permutation(namp)   # the two denoised tables are ordered independently
    seqsF = [seqsF[i] for i in permF]
    seqsR = [seqsR[i] for i in permR]
    posF = {int(old) + 1: new + 1 for new, old in enumerate(permF)}
    posR = {int(old) + 1: new + 1 for new, old in enumerate(permR)}
    w = rng.dirichlet(np.full(namp, 0.5))
    fwd, rev = [], []
    for _ in range(npairs):
        a = int(rng.choice(namp, p=w)) + 1
        f, r = posF[a], posR[a]
        u = rng.random()
        if u < 0.06:
            r = int(rng.integers(1, namp + 1))             # chimeric / mispaired read pair
        elif u < 0.09:
            f = -1                                         # NA: the forward read was not assigned
        elif u < 0.12:
            r = -1
        fwd.append(f)
        rev.append(r)
    n0F = rng.integers(1, 200, size=namp).astype(np.int32)
    n0R = rng.integers(1, 200, size=namp).astype(np.int32)
    return dict(seqsF=seqsF, seqsR=seqsR, fwd=np.array(fwd, dtype=np.int32), rev=np.array(rev, dtype=np.int32), n0F=n0F, n0R=n0R)
