"""oracle/merge.py — TEST INFRASTRUCTURE ONLY.

CPU restatement of mergePairs() (/root/reference/R/paired.R:92-201) for ONE sample: the R-level bookkeeping in plain
Python (small cases), the alignment and the two C helpers from a checker module (`oracle.ref`: the reference's own
evaluate.cpp compiled in place; or `oracle.cport`: the plain-C restatement).  Rows come back in the reference's order:
unique (forward, reverse) pairs in order of first appearance, stably sorted by decreasing abundance (:182)."""
import numpy as np

_COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    """R/misc.R rc(): reverse complement."""
    return s.translate(_COMP)[::-1]


def merge_pairs(fwd, rev, seqsF, n0F, seqsR, n0R, checker, min_overlap=12, max_mismatch=0, trim_overhang=False,
                just_concatenate=False, return_rejects=False):
    """fwd / rev: per read pair, the 1-based index of its denoised forward / reverse sequence (dadaF$map[derepF$map],
    paired.R:119-120) or a negative value / None for NA.  Returns a list of dict rows."""
    first, count = {}, {}
    for f, r in zip(fwd, rev):                              # unique(pairdf) keeps first appearances (:123)
        key = (int(f) if f is not None and f > 0 else None, int(r) if r is not None and r > 0 else None)
        if key not in first:
            first[key] = len(first)
        count[key] = count.get(key, 0) + 1
    ups = [k for k in first if k[0] is not None and k[1] is not None]   # (:124-125)
    rows = []
    for f, r in ups:
        F, R = seqsF[f - 1], rc(seqsR[r - 1])
        row = {"forward": f, "reverse": r, "abundance": count[(f, r)]}
        if just_concatenate:                                # (:139-147)
            row.update(sequence=F + "NNNNNNNNNN" + R, nmatch=0, nmismatch=0, nindel=0, prefer=None, accept=True)
        else:
            sc = (1, -64, -64) if max_mismatch == 0 else (1, -8, -8)    # (:152-157)
            a1, a2 = checker.C_nwalign(F, R, sc[0], sc[1], sc[2], None, -1, True) if hasattr(checker, "C_nwalign") else \
                checker.nwalign(F, R, sc[0], sc[1], sc[2], band=-1)
            m, mm, ind = checker.eval_pair(a1, a2)
            prefer = 1 + int(n0R[r - 1] > n0F[f - 1])        # (:164)
            accept = (m >= min_overlap) and ((mm + ind) <= max_mismatch)
            seq = checker.pair_consensus(a1, a2, prefer, trim_overhang)
            row.update(sequence=seq if accept else "", nmatch=m, nmismatch=mm, nindel=ind, prefer=prefer, accept=accept)
        rows.append(row)
    order = np.argsort(-np.array([r["abundance"] for r in rows], dtype=np.int64), kind="stable") if rows else []
    rows = [rows[i] for i in order]
    if not return_rejects:
        rows = [r for r in rows if r["accept"]]
    return rows
