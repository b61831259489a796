"""Seeded synthetic amplicon samples for the parity tests and bench.py.

The reference ships no benchmark inputs; BASELINE.json's configs are defined on synthetic
uniques.  The generator follows the survey's recipe: families of true variants derived
from one ancestor, Zipf abundances, a linear quality ramp with Gaussian jitter, and
substitution errors drawn from the error matrix itself (so the data are self-consistent
with the model), then dereplication exactly as ``derepFastq`` would do it
(dada2_amd/io.py).  Everything is numpy-vectorised so 10^6 uniques take ~a minute.
"""
from __future__ import annotations

import numpy as np

from .io import Derep

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def true_variants(rng, G: int, L: int, Lmin: int = None, Lmin5: int = None):
    """G/8 roots at divergence U(0.03,0.25) from one ancestor; each root + 7 variants at
    Hamming 1..7 from it.  Returns uint8 codes 0..3, shape [G, L] and lengths [G].
    ``Lmin``: 3'-ragged variants (lengths U{Lmin..L}, the tail cut off).  ``Lmin5``: 5'-ragged variants (a variant is
    the SUFFIX of its full-length sequence, start offset U{0..L-Lmin5}) — primer-trim variation, ordinary for long
    amplicons: the alignment of such a read to a full-length centre runs |i - j| = offset cells off the main diagonal."""
    nroot = max(1, G // 8)
    anc = rng.integers(0, 4, size=L, dtype=np.uint8)
    out = np.empty((nroot * 8, L), dtype=np.uint8)
    for r in range(nroot):
        root = anc.copy()
        div = rng.uniform(0.03, 0.25)
        pos = rng.choice(L, size=max(1, int(round(div * L))), replace=False)
        root[pos] = (root[pos] + rng.integers(1, 4, size=pos.size, dtype=np.uint8)) & 3
        out[r * 8] = root
        for h in range(1, 8):
            v = root.copy()
            p = rng.choice(L, size=h, replace=False)
            v[p] = (v[p] + rng.integers(1, 4, size=h, dtype=np.uint8)) & 3
            out[r * 8 + h] = v
    lens = np.full(nroot * 8, L, dtype=np.int32)
    if Lmin is not None and Lmin < L:
        lens = rng.integers(Lmin, L + 1, size=nroot * 8).astype(np.int32)
    if Lmin5 is not None and Lmin5 < L:
        off = rng.integers(0, L - Lmin5 + 1, size=nroot * 8)
        off[::8] = 0                                  # the roots stay full length at the 5' end
        for g in range(nroot * 8):
            o = int(min(off[g], max(0, lens[g] - max(8, Lmin5 // 4))))
            if o > 0:
                out[g, : L - o] = out[g, o:].copy()
                lens[g] -= o
    return out[:G], lens[:G]


def _derep_codes(codes: np.ndarray, lens: np.ndarray, quals: np.ndarray) -> Derep:
    """Dereplicate reads given as code rows (0..3, padded with 255 past each read's end)."""
    n, L = codes.shape
    key = np.ascontiguousarray(codes).view(np.dtype((np.void, L))).ravel()
    uniq, first, inv, counts = np.unique(key, return_index=True, return_inverse=True, return_counts=True)
    # np.unique sorts the void rows bytewise; with codes 0..3 == A<C<G<T and 255 padding a shorter
    # read sorts AFTER its extensions, whereas C-locale strings sort a prefix first.  Re-rank by
    # the actual strings only when lengths vary.
    U = uniq.size
    ulen = lens[first]
    # per-unique quality sums: singletons are their own row; the rest via sort + reduceat
    mean = quals[first].astype(np.float64)
    multi = counts > 1
    if multi.any():
        sel = np.nonzero(multi[inv])[0]
        o = sel[np.argsort(inv[sel], kind="stable")]
        starts = np.concatenate([[0], np.cumsum(counts[multi])[:-1]])
        qs = np.add.reduceat(quals[o].astype(np.int32), starts, axis=0)
        mean[multi] = qs.astype(np.float64) / counts[multi][:, None]
    ucodes = codes[first]
    lut = np.full(256, ord("-"), dtype=np.uint8)
    lut[:4] = _ACGT
    asc = lut[ucodes]
    seqs = [asc[u, : ulen[u]].tobytes().decode("ascii") for u in range(U)]
    if (ulen != L).any():
        lex = np.array(sorted(range(U), key=lambda u: seqs[u]), dtype=np.int64)
    else:
        lex = np.arange(U)
    order = lex[np.argsort(-counts[lex], kind="stable")]   # (-abundance, sequence): stable R order()
    mean = mean[order]
    for k, u in enumerate(order):
        mean[k, ulen[u]:] = np.nan
    rank = np.empty(U, dtype=np.int64)
    rank[order] = np.arange(U)
    return Derep([seqs[u] for u in order], counts[order].astype(np.int32), mean, rank[inv].astype(np.int32))


def make_sample(err: np.ndarray, n_uniques: int, L: int = 250, G: int = 256, seed: int = 0, Lmin: int = None,
                q_hi: float = 38.0, q_lo: float = 22.0, q_sd: float = 4.0, q_max: int = 40, indel_rate: float = 0.0,
                zipf: float = 1.1, variants=None, chunk: int = 200_000, Lmin5: int = None, ins_rate: float = 0.0) -> Derep:
    """Draw reads until the dereplicated unique count reaches ``n_uniques`` (then trim the
    rarest uniques so N is exact), and dereplicate.  ``err`` is the 16 x Q matrix errors are
    drawn from.  ``variants`` (codes, lens) may be passed to share truth across samples."""
    rng = np.random.default_rng(seed)
    tv, tl = variants if variants is not None else true_variants(rng, G, L, Lmin, Lmin5)
    G = tv.shape[0]
    w = (np.arange(1, G + 1, dtype=np.float64)) ** (-zipf)
    w /= w.sum()
    err = np.asarray(err, dtype=np.float64)
    Q = err.shape[1]
    ramp = np.linspace(q_hi, q_lo, L)
    all_codes, all_lens, all_q = [], [], []
    seen = 0
    seen_keys = set()
    while True:
        g = rng.choice(G, size=chunk, p=w)
        codes = tv[g].copy()
        lens = tl[g].copy()
        q = np.clip(np.rint(ramp[None, :] + rng.normal(0.0, q_sd, size=(chunk, L))), 2, min(q_max, Q - 1)).astype(np.uint8)
        # substitution errors: P(b -> b') = err[4b+b', q]
        u = rng.random(size=(chunk, L))
        b = codes.astype(np.int64)
        cum = np.zeros((chunk, L))
        newc = codes.copy()
        done = np.zeros((chunk, L), dtype=bool)
        for k in range(1, 4):
            tgt = (b + k) & 3
            cum += err[4 * b + tgt, q]
            hit = (~done) & (u < cum)
            newc[hit] = tgt[hit].astype(np.uint8)
            done |= hit
        codes = newc
        if indel_rate > 0:
            # rare single-base deletions (exercise the band); applied per read, at most one
            has = rng.random(chunk) < indel_rate * lens
            for r in np.nonzero(has)[0]:
                p = int(rng.integers(1, lens[r] - 1))
                codes[r, p:-1] = codes[r, p + 1:]
                q[r, p:-1] = q[r, p + 1:]
                lens[r] -= 1
        if ins_rate > 0:
            # rare single-base insertions (a read at full length keeps its length: its last base falls off)
            has = rng.random(chunk) < ins_rate * lens
            for r in np.nonzero(has)[0]:
                p = int(rng.integers(1, lens[r] - 1))
                codes[r, p + 1:] = codes[r, p:-1].copy()
                q[r, p + 1:] = q[r, p:-1].copy()
                codes[r, p] = rng.integers(0, 4)
                lens[r] = min(L, lens[r] + 1)
        pad = np.arange(L)[None, :] >= lens[:, None]
        codes[pad] = 255
        q[pad] = 0
        all_codes.append(codes); all_lens.append(lens); all_q.append(q)
        keys = np.ascontiguousarray(codes).view(np.dtype((np.void, L))).ravel()
        seen_keys.update(np.unique(keys).tolist())
        seen = len(seen_keys)
        if seen >= n_uniques:
            break
    codes = np.concatenate(all_codes); lens = np.concatenate(all_lens); q = np.concatenate(all_q)
    d = _derep_codes(codes, lens, q)
    if d.nraw > n_uniques:  # drop the rarest (last) uniques and the reads that map to them
        keep = n_uniques
        d = Derep(d.seqs[:keep], d.abundances[:keep], d.quals[:keep], np.where(d.map < keep, d.map, -1).astype(np.int32))
    maxlen = max(len(s) for s in d.seqs)   # quals has exactly maxlen columns (Rmain.cpp:69-72)
    if d.quals.shape[1] != maxlen:
        d = Derep(d.seqs, d.abundances, np.ascontiguousarray(d.quals[:, :maxlen]), d.map)
    return d
