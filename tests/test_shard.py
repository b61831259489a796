"""One sample over several ranks (dada2hip_sample_run_sharded / dada2_amd.shard; SURVEY.md §8e last row, §8f rank 4).

CPU (this file's unmarked tests): two and three ranks under torch.distributed's gloo backend, each rank running the REAL
kernels and driver of the library through the functional emulator of tests/emu, every rank's result compared bit for bit
with the unsharded run.  GPU (-m gpu): the same with the real library - two gloo ranks sharing the box's one GPU, and the
RCCL path at world size 1."""
import os
import shutil
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "shard_worker.py")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def run_group(which, backend, case, world, timeout=900, env_extra=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, WORKER, which, backend, case], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r}/{world} ok" in out, f"rank {r}:\n" + out[-3000:]
    return outs


needs_emu = pytest.mark.skipif(not (os.path.exists(CXX) or shutil.which(CXX)), reason="no host clang++ for the emulator build")


@needs_emu
@pytest.mark.parametrize("world,case", [(2, "sam1F_default"), (3, "sam1F_priors"), (2, "synth:1500:120:16")])
def test_sharded_run_under_gloo_equals_the_single_process_result(world, case):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build as emu_build
    emu_build.build()            # once, before the ranks race for it
    run_group("emu", "gloo", case, world)


@needs_emu
def test_a_failing_rank_takes_the_others_down_with_an_error_not_a_hang():
    """A rank that fails between exchange points says so at the next one (the 8-byte size gather every exchange opens with
    carries -1): its peers return DADA2HIP_ERR_RUNTIME there instead of blocking in a collective."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build as emu_build
    emu_build.build()
    outs = run_group("emu", "gloo", "sam1F_default", 2, timeout=300, env_extra={"SHARD_TEST_FAIL": "1"})
    assert "injected failure on rank 1" in outs[1] and "another rank of the sharded run failed" in outs[0], outs


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [(2, "sam1F_default"), (2, "synth:20000:250:96"), (3, "synth3000_default")])
def test_gpu_two_ranks_share_one_gpu_under_gloo(world, case):
    run_group("hip", "gloo", case, world)


@pytest.mark.gpu
def test_gpu_sharded_path_under_rccl_world_size_one():
    run_group("hip", "nccl", "sam2F_nogreedy", 1)
