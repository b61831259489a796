"""Worker of tests/test_shard.py: one rank of a torch.distributed group running dada2_amd.shard.dada_sharded on a seeded
sample and checking the result against the unsharded run of the same library in the same process.
usage: python shard_worker.py <lib: 'emu' | 'hip'> <backend> <case>   (RANK / WORLD_SIZE / MASTER_* from the environment)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
which, backend, case = sys.argv[1], sys.argv[2], sys.argv[3]
import torch  # noqa: E402  (before the library: conftest.py's note on the two HIP runtimes)
import torch.distributed as dist  # noqa: E402
from dada2_amd import _lib  # noqa: E402
if which == "emu":   # the functional emulator of tests/emu: the real kernels and driver on the CPU (test infrastructure)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build as emu_build
    _lib.LIB_PATH = emu_build.build()
import numpy as np  # noqa: E402
from helpers import assert_results_equal, case_inputs, tperr1  # noqa: E402
from dada2_amd import api  # noqa: E402
from dada2_amd.opts import DadaOpts  # noqa: E402
from dada2_amd.shard import dada_sharded  # noqa: E402
from dada2_amd.synth import make_sample  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    if case.startswith("synth:"):
        _, n, L, G = case.split(":")
        d = make_sample(tperr1(), int(n), L=int(L), G=int(G), seed=4242, chunk=max(4000, int(n)))
        err, pri, o = tperr1(), None, DadaOpts()
    else:
        d, err, pri, o, exp, meta = case_inputs(case)
    smp = api.Sample.from_derep(d, pri, device=0)
    if os.environ.get("SHARD_TEST_FAIL"):
        # one rank fails in the middle of the run (here: at the opening of its 6th exchange point): EVERY rank must come back
        # with an error instead of waiting for it in a collective for ever
        from dada2_amd.shard import make_exchange
        ex0 = make_exchange(dist, None)
        state = {"n": 0, "fired": False}

        def ex(kind, send, recv):
            if rank == 1 and not state["fired"] and kind == 0 and len(send) == 8:
                state["n"] += 1
                if state["n"] == 6:
                    state["fired"] = True
                    raise RuntimeError("injected failure on rank 1")
            return ex0(kind, send, recv)
        try:
            smp.run_sharded(err, o, rank, world, ex)
            print(f"rank {rank}/{world} UNEXPECTEDLY finished", flush=True)
        except Exception as e:   # noqa: BLE001
            print(f"rank {rank}/{world} ok: failed as it should: {e}", flush=True)
        smp.close()
        sys.exit(0)
    want = smp.run(err, o)
    got = dada_sharded(smp, err, o, dist=dist, collective_device=torch.device("cuda", 0) if backend == "nccl" else None)
    smp.close()
    assert_results_equal(got, want, exact_float=True, check_birth_from=pri is None)
    # the reference's own work counters (dada.h:113-114), summed over the blocks
    assert got.stats["ncompare"] - got.stats["nskipped"] == want.stats["ncompare"] - want.stats["nskipped"]
    assert got.stats["nshroud"] == want.stats["nshroud"]
    print(f"rank {rank}/{world} ok: {got.nclust} partitions, collectives {got.stats['shard_collectives']}", flush=True)
finally:
    dist.destroy_process_group()
