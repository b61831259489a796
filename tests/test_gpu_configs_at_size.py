"""BASELINE.json's configurations at their STATED sizes, every output against the reference's own C++ (oracle/_ref on all
host cores).  The 1 M-unique fixed-error call (configs[2]'s size) and the 100 k call (configs[1]) live in
test_gpu_scale_and_edges.py / test_gpu_parity.py; here:

  configs[2]  selfConsist=TRUE learnErrors loop (R/dada.R:256-405): EVERY pass of the loop - the all-ones / MAX_CLUST=1
              start (R/dada.R:298), the resident-state reset between passes, each refit (R/errorModels.R:462-471 +
              noqualErrfun) - at 100 k uniques, where the class cache, store growth and multi-batch planning are active
  configs[3]  8 samples x 250 k uniques through dada2hip_run_multi AND through multi.dada_multi (nccl, world size 1)
  configs[4]  200 k uniques of ~1 500 nt, BAND_SIZE 32, MAX_CLUST 32: the wide kernel's HBM pointer ring, store growth
              and > 16 partitions at 1.5 kb

Everything is skipped where the prebuilt reference is absent.  (-k "not at_size" deselects the module.)"""
import os

import numpy as np
import pytest

from helpers import P_RTOL, assert_results_equal, tperr1
from dada2_amd.io import extend_err
from dada2_amd.opts import DadaOpts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dada2_amd import api as a
    return a


@pytest.fixture(scope="module")
def ref():
    from oracle import ref as r
    if not r.available():
        pytest.skip("oracle/_ref not present")
    r.set_threads(os.cpu_count() or 1)
    yield r
    r.set_threads(1)


def test_at_size_selfconsist_every_pass_vs_reference(api, ref):
    from dada2_amd.synth import make_sample
    d = make_sample(tperr1(), 100_000, L=250, G=256, seed=20260925 + 2)
    o = DadaOpts(OMEGA_C=0, MAX_CONSIST=6)          # learnErrors() calls dada(selfConsist=TRUE, OMEGA_C=0, ...) (R/errorModels.R:334,360-361)
    passes = []
    res, err_out, errs = api.dada(d, None, self_consist=True, opts=o,
                                  on_pass=lambda k, used, mc, rs: passes.append((k, used[0].copy(), mc, rs[0])))
    assert len(passes) >= 3 and passes[0][2] == 1 and np.all(passes[0][1] == 1.0)     # R/dada.R:298 start
    assert len(errs) == len(passes) - 1
    ncl = []
    for k, err_used, max_clust, got in passes:
        want = ref.dada_uniques(d.seqs, d.abundances, None, err_used, d.quals, o, max_clust=max_clust, multithread=True)
        assert got.nclust == want.nclust, (k, got.nclust, want.nclust)
        assert_results_equal(got, want, p_rtol=P_RTOL)
        ncl.append(got.nclust)
    assert ncl[0] == 1 and max(ncl) > 50
    # the refit the loop applied between the passes is the one the counts of the previous pass give (accumulateTrans + noqual)
    for (k0, _, _, r0), (k1, e1, _, _) in zip(passes[:-1], passes[1:]):
        want_err = api.noqual_errfun(api.accumulate_trans([r0.subqual]))
        if k0 == 0:
            want_err[[0, 5, 10, 15], :] = 1.0                                       # R/dada.R:385-388
        assert np.array_equal(extend_err(want_err, int(np.ceil(np.nanmax(d.quals)))), e1), k1
    assert_results_equal(res, passes[-1][3], exact_float=True)


def test_at_size_config4_eight_samples_through_run_multi_and_dada_multi(api, ref):
    import socket
    import torch
    import torch.distributed as dist
    import at_size
    from dada2_amd.multi import dada_multi
    dereps, err, o, wants = at_size.get("cfg4")    # bench.py --config 4's pool (8 samples x 250 k uniques) + the reference's results
    assert len(dereps) == 8 and all(d.nraw == 250_000 for d in dereps)
    got_multi = api.dada_uniques_multi(dereps, err, o, devices=(0, 0))   # C entry: two host threads share this box's one GPU
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got_dist, err_out, _ = dada_multi(dereps, err, self_consist=False, opts=o, dist=dist, device=torch.device("cuda", 0))
    finally:
        dist.destroy_process_group()
    assert sorted(got_dist) == list(range(8))
    trans = np.zeros((16, 41), dtype=np.int64)
    for i, want in enumerate(wants):
        assert want.nclust > 100
        assert_results_equal(got_multi[i], want, p_rtol=P_RTOL)
        assert_results_equal(got_dist[i], want, p_rtol=P_RTOL)
        trans += want.subqual
    # the loop's only exchange: the all-reduced transition counts and the refit from them equal the reference's sum
    assert np.array_equal(err_out, api.noqual_errfun(trans))


def test_at_size_config5_long_reads_200k_vs_reference(api, ref):
    import at_size
    dereps, err, o, wants = at_size.get("cfg5")    # bench.py --config 5's sample (200 k uniques of 1 450-1 510 nt) + the reference's result
    d, want = dereps[0], wants[0]
    assert d.nraw == 200_000 and o.BAND_SIZE == 32 and o.MAX_CLUST == 32
    got = api.dada_uniques(d.seqs, d.abundances, None, err, d.quals, o)
    assert got.nclust == want.nclust == 32
    assert_results_equal(got, want, p_rtol=P_RTOL)
