"""One sample over several GPUs: the ranks of a ``torch.distributed`` group each do one block of the uniques
(``dada2hip_sample_run_sharded``, include/dada2hip.h).

The reference runs a sample as one serial ``dada_uniques`` call (R/dada.R:266) - it has no counterpart of this.  Here
every rank holds the whole sample (sequences, qualities, k-mer records: the inputs are small next to the work) and the
library's host-driven round loop exchanges, through the callback below, exactly what the reference's serial bookkeeping
shares between uniques: the moves of each ``b_shuffle2`` call (src/cluster.cpp:242-259), the candidates of each ``b_bud``
(:284-308), and the per-unique results at the end.  Every rank returns the same, complete ``DadaResult`` - bit-identical
to the one-GPU result (tests/test_shard.py).

Collectives (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests):
  kind 0  ``all_gather`` of equal-size byte payloads (the library pads variable-size lists itself),
  kind 1  ``all_reduce(SUM)`` of int64.
They are small and latency-bound: one to three per shuffle call, one or two per birth."""
from __future__ import annotations

import numpy as np

from .opts import DadaOpts


def make_exchange(dist, device=None):
    """The ``exchange(kind, send, recv)`` callback of ``Sample.run_sharded`` over a torch.distributed group.
    ``device``: torch device the collective tensors live on (a cuda device for nccl; None = CPU for gloo)."""
    import torch
    world = dist.get_world_size()
    counters = {"all_gather": 0, "all_reduce": 0, "bytes": 0}

    def exchange(kind, send, recv):
        n = len(send)
        if n == 0:
            return
        if kind == 0:
            src = torch.frombuffer(bytearray(send), dtype=torch.uint8)
            if device is not None:
                src = src.to(device)
            out = torch.empty(world * n, dtype=torch.uint8, device=src.device)
            dist.all_gather_into_tensor(out, src)
            recv[:] = out.cpu().numpy().tobytes()
            counters["all_gather"] += 1
        elif kind == 1:
            t = torch.frombuffer(bytearray(send), dtype=torch.int64)
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            recv[:] = t.cpu().numpy().tobytes()
            counters["all_reduce"] += 1
        else:
            raise ValueError(f"unknown exchange kind {kind}")
        counters["bytes"] += n
    exchange.counters = counters
    return exchange


def dada_sharded(derep_or_sample, err, opts: DadaOpts = None, *, dist, device_index: int = 0, collective_device=None, max_clust=None):
    """``dada_uniques`` of ONE sample with the per-unique work split over the ranks of ``dist``.
    ``derep_or_sample``: a ``Derep`` (made resident here) or an ``api.Sample`` already resident on this rank's GPU."""
    from . import api
    o = (opts or DadaOpts()).normalised()
    own = not isinstance(derep_or_sample, api.Sample)
    smp = api.Sample.from_derep(derep_or_sample, device=device_index) if own else derep_or_sample
    try:
        ex = make_exchange(dist, collective_device)
        res = smp.run_sharded(err, o, dist.get_rank(), dist.get_world_size(), ex, max_clust=max_clust)
        res.stats["shard_collectives"] = dict(ex.counters)
        return res
    finally:
        if own:
            smp.close()
