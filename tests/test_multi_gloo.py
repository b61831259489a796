"""N > 1 path on CPU: world_size-2 gloo.  The per-sample runner is the oracle here (no GPU in this
container); what is under test is the sharding, the all-reduce of the 16 x Q transition matrix
(accumulateTrans, R/errorModels.R:462-471) and that every rank refits the same error matrix and
stops at the same pass — i.e. the result equals the single-process loop over all samples."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import tperr1
    from dada2_amd.multi import dada_multi
    from dada2_amd.opts import DadaOpts
    from dada2_amd.synth import make_sample
    from oracle import cport

    class OracleRunner:
        def __init__(self, d): self.d = d
        def run(self, err, opts, max_clust=None):
            return cport.dada_uniques(self.d.seqs, self.d.abundances, None, err, self.d.quals, opts, max_clust=max_clust)
        def close(self): pass

    dereps = [make_sample(tperr1(), 400, L=100, G=8, seed=100 + i, chunk=2000) for i in range(5)]
    res, err, errs = dada_multi(dereps, None, self_consist=True, opts=DadaOpts(OMEGA_C=0, MAX_CONSIST=3),
                                make_runner=OracleRunner, dist=dist)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), err=err, nerrs=len(errs), idx=np.array(sorted(res)),
             nclust=np.array([res[i].nclust for i in sorted(res)]),
             trans=np.stack([np.pad(res[i].subqual, ((0, 0), (0, 41 - res[i].subqual.shape[1]))) for i in sorted(res)]))
    dist.destroy_process_group()


def test_two_rank_sharded_selfconsist_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["err"], r1["err"]) and int(r0["nerrs"]) == int(r1["nerrs"])
    assert r0["idx"].tolist() == [0, 2, 4] and r1["idx"].tolist() == [1, 3]
    # single-process reference of the same loop
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import tperr1
    from dada2_amd.multi import dada_multi
    from dada2_amd.opts import DadaOpts
    from dada2_amd.synth import make_sample
    from oracle import cport

    class OracleRunner:
        def __init__(self, d): self.d = d
        def run(self, err, opts, max_clust=None):
            return cport.dada_uniques(self.d.seqs, self.d.abundances, None, err, self.d.quals, opts, max_clust=max_clust)
        def close(self): pass

    dereps = [make_sample(tperr1(), 400, L=100, G=8, seed=100 + i, chunk=2000) for i in range(5)]
    res, err, errs = dada_multi(dereps, None, self_consist=True, opts=DadaOpts(OMEGA_C=0, MAX_CONSIST=3), make_runner=OracleRunner)
    assert np.array_equal(err, r0["err"]) and len(errs) == int(r0["nerrs"])
    allnc = {int(i): int(n) for r in (r0, r1) for i, n in zip(r["idx"], r["nclust"])}
    assert [allnc[i] for i in range(5)] == [res[i].nclust for i in range(5)]


def test_shard_round_robin():
    from dada2_amd.multi import shard
    assert shard(8, 0, 8) == [0] and shard(8, 3, 4) == [3, 7] and shard(5, 1, 2) == [1, 3]
    assert sorted(sum((shard(11, r, 4) for r in range(4)), [])) == list(range(11))
    # unequal samples: longest first, each to the least loaded rank; every rank derives the same deal
    sizes = [100, 900, 300, 300, 50, 800, 20, 10]
    deal = [shard(8, r, 3, sizes) for r in range(3)]
    assert sorted(sum(deal, [])) == list(range(8)) and deal[0] == [1] and deal[1] == [5]
    assert shard(8, 2, 4, [7] * 8) == [2, 6]
