"""Sample-level sharding of a multi-sample dada()/learnErrors run over the GPUs of one node.

The path shards naturally at sample granularity: each ``dada_uniques`` call
depends only on its own sample and the shared error matrix (R/dada.R:266-366).  One process per
GPU (``torch.distributed``, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests); the samples are dealt longest-first
(``shard``: r, r+W, r+2W, ... when they are of one size) and stay resident on their rank's GPU for the whole selfConsist loop.  The only
cross-rank exchange is ``accumulateTrans`` (R/errorModels.R:462-471): one all-reduce(sum) of the
16 x Q int64 transition-count matrix per pass (<= 12 KB, latency-bound; xGMI bandwidth is
irrelevant at this size).  Every rank then refits ``err`` from the identical reduced counts, so no
broadcast is needed and all ranks take the same termination decision.
"""
from __future__ import annotations

import numpy as np

from .io import extend_err
from .opts import DadaOpts


def shard(n_samples: int, rank: int, world: int, sizes=None):
    """Indices of the samples rank ``rank`` owns.  With ``sizes`` (uniques per sample) the deal is longest-first:
    samples in decreasing cost - a sample costs about N x partitions, and partitions grow roughly as sqrt(N) on these data -
    each to the rank with the least work so far; every rank computes the same deal from the same sizes.  Equal sizes (or no
    sizes) give the round-robin r, r + W, r + 2 W, ..."""
    if sizes is None or len(set(int(x) for x in sizes)) <= 1:
        return list(range(rank, n_samples, world))
    cost = [float(n) ** 1.5 for n in sizes]
    order = sorted(range(n_samples), key=lambda i: (-cost[i], i))
    load = [0.0] * world
    owner = [0] * n_samples
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += cost[i]
    return [i for i in range(n_samples) if owner[i] == rank]


def allreduce_trans(local_trans: np.ndarray, maxcol: int, dist=None, device=None) -> np.ndarray:
    """Sum of the per-rank 16 x Q transition matrices (accumulateTrans across ranks)."""
    buf = np.zeros((16, maxcol), dtype=np.int64)
    buf[:, : local_trans.shape[1]] += local_trans
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return buf
    import torch
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def dada_multi(dereps, err, *, self_consist=False, err_fun=None, opts: DadaOpts = None, make_runner=None, dist=None,
               device=None, max_col: int = None, on_pass=None, inflight: int = None):
    """dada() over many samples, sharded across the ranks of ``dist``.

    ``dereps``       all samples (every rank sees the list; only its shard is touched)
    ``make_runner``  derep -> object with .run(err, opts, max_clust=...) -> DadaResult and .close();
                     default = GPU-resident dada2_amd.api.Sample on this rank's device
    ``inflight``     samples of this rank running at a time (1 = the serial loop of R/dada.R:266).  Default: 3 with the library's
                     own runners (their ``run`` is thread-safe: one resident sample each; up to three hold a persistent slot of the
                     device side by side while their blocks fit three quarters of its CUs, else they take turns - driver.cpp
                     SlotSem; each holds its own round buffers, so device memory per rank grows with it), 1 with user-supplied
                     ``make_runner`` objects, whose ``run`` is then never called from two threads at once
    Returns (dict sample_index -> DadaResult for the local shard, err_out, list of err tried).
    Mirrors the loop of R/dada.R:256-405 (see dada2_amd.api.dada for the single-process form)."""
    from .api import accumulate_trans, noqual_errfun
    o = (opts or DadaOpts()).normalised()
    err_fun = err_fun or noqual_errfun
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    mine = shard(len(dereps), rank, world, sizes=[d.nraw for d in dereps])
    if inflight is None:
        inflight = 3 if make_runner is None else 1
    if make_runner is None:
        from .api import Sample
        dev_index = device.index if device is not None and getattr(device, "index", None) is not None else 0
        make_runner = lambda d: Sample.from_derep(d, device=dev_index)   # noqa: E731
    runners = {i: make_runner(dereps[i]) for i in mine}
    qmax_all = max(d.qmax() for d in dereps)
    maxcol = max_col or max(41, qmax_all + 1, 0 if err is None else np.asarray(err).shape[1])
    initialize = self_consist and err is None
    nconsist = 0 if initialize else 1
    errs, results = [], {}
    try:
        while True:
            if nconsist > 0:
                errs.append(np.array(err, copy=True))
            local = np.zeros((16, maxcol), dtype=np.int64)
            used = {}
            def one(i):
                d = dereps[i]
                qmax = d.qmax()
                erri = np.ones((16, max(41, qmax + 1))) if initialize else extend_err(err, qmax)
                return i, erri, runners[i].run(erri, o, max_clust=1 if initialize else None)
            # a rank's samples `inflight` at a time (threads; the library call releases the GIL): the rounds of one sample fill the
            # GPU and take turns, the round 0, final passes and result marshalling of the others run beside them
            if inflight > 1 and len(mine) > 1:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=min(inflight, len(mine))) as pool:
                    done = list(pool.map(one, mine))
            else:
                done = [one(i) for i in mine]
            for i, erri, res in done:
                results[i] = res
                used[i] = erri
                t = res.subqual
                local[:, : t.shape[1]] += t
            if on_pass is not None:   # (tests: every pass of the loop is checked, not just the last)
                on_pass(0 if initialize else len(errs), used, 1 if initialize else None, dict(results))
            cur = allreduce_trans(local, maxcol, dist, device)
            new_err = err_fun(cur)
            if initialize:
                initialize = False
                new_err[[0, 5, 10, 15], :] = 1.0
            err = new_err
            if (not self_consist) or any(np.array_equal(e, err) for e in errs) or nconsist >= o.MAX_CONSIST:
                break
            nconsist += 1
    finally:
        for r in runners.values():
            if hasattr(r, "close"):
                r.close()
    return results, err, errs
