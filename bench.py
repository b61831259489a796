#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: unique reads denoised / s of dada() wall-clock.

One "step" = ONE BOUNDARY CALL `dada2hip_dada_uniques` (include/dada2hip.h) on host inputs — the marshalling of
the R-side vectors, H2D upload, k-mer build, every divisive round, the final alignments, the output tables and
the D2H of the six result objects — exactly what SURVEY.md §8(d) defines as the metric's wall clock ("inputs
already on host, includes H2D, all kernels, D2H of results").  Workload by --config (SURVEY.md §8 numbering):

  3 (default)  1 000 000 unique 250-nt synthetic reads, tperr1 fixed error matrix   <- the headline (BASELINE.json)
               (--selfconsist runs configs[2]'s learnErrors-style loop on it: err from all-ones, noqual refit)
  2            100 000 unique 250-nt reads                                          (BASELINE.json configs[1])
  4            8 samples x 250 000 uniques, samples sharded round-robin over the ranks (strong scaling, configs[3])
  --shard      (configs 2/3/5) ONE sample whose uniques are split over the ranks: every rank does the comparisons, shuffles and
               p-values of its block, the moves / bud candidates / final sums are exchanged over RCCL (strong scaling)
  5            200 000 unique ~1 500-nt reads, BAND_SIZE 32, 94 quality columns     (configs[4])

With --gpus N (configs 2/3/5) every rank denoises its own sample of the same size (weak scaling: the path shards
at sample granularity); the only collective is the RCCL all-reduce of the 16 x Q transition-count
matrix (accumulateTrans, R/errorModels.R:462-471), inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with, besides the contract's keys:
  roofline        dominant kernel of a fully event-timed pass (DADA2HIP_PROFILE=1: every launch timed, nothing
                  extrapolated): algorithmic work / measured duration vs the guide's peak and the measured peak
  cpu_baseline    the reference's own C++ (oracle/_ref: -O2 as R builds it, and -O3) on this box's host cores
  resident        the same pass on a sample already resident in HBM (what selfConsist passes 2..n cost)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# (before anything initialises the HIP runtime - torch does, in main(): the library wants eight hardware queues for its streams,
#  dada2_amd/csrc/knobs.h knobs_process_defaults; a caller's own setting wins)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy); integer VALU = 256 CU x 4 SIMD-32 x
# 2.4 GHz = 78.6 T 32-bit lane-ops/s (the FP32 vector rate, 157.3 TFLOPS, counts an FMA as 2)
PEAK_HBM_GBS = 8000.0
PEAK_VALU_TOPS = 78.6
INT_OPS_PER_CELL = 8        # fixed algorithmic constant, SURVEY.md §8d

CONFIGS = {
    2: dict(uniques=100_000, length=250, variants=256, band=16, samples=1),
    3: dict(uniques=1_000_000, length=250, variants=2048, band=16, samples=1),
    4: dict(uniques=250_000, length=250, variants=512, band=16, samples=8),
    5: dict(uniques=200_000, length=1510, variants=128, band=32, samples=1, lmin=1450, q_hi=93.0, q_lo=30.0, q_max=93,
            indel=1e-4),
}


def measured_peaks():
    """Peaks pinned by tools/microbench on the GPU box (profiles/peaks_*.json), if a record is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "peaks_*.json")))
    if not files:
        return None
    try:
        p = json.load(open(files[-1]))
        p["_file"] = os.path.relpath(files[-1], ROOT)
        return p
    except Exception:
        return None


def _cached(make_sample):
    """Synthetic samples are seeded and deterministic but take minutes to draw at 10^6 uniques / 1.5 kb reads: keep them in a
    scratch directory (DADA2HIP_BENCH_CACHE, default /tmp/dada2hip_bench_cache; empty string = off) so that the several
    bench / profiler invocations of one GPU-box session draw each sample once.  Inputs only - nothing computed is cached."""
    import hashlib
    from dada2_amd.io import Derep
    cdir = os.environ.get("DADA2HIP_BENCH_CACHE", "/tmp/dada2hip_bench_cache")

    def wrapped(err, n, **kw):
        if not cdir:
            return make_sample(err, n, **kw)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(err).tobytes())
        h.update(repr((n, sorted((k, (v[0].tobytes(), v[1].tobytes()) if k == "variants" and v is not None else v) for k, v in kw.items()))).encode())
        path = os.path.join(cdir, h.hexdigest()[:24] + ".npz")
        try:
            if os.path.exists(path):
                z = np.load(path)
                return Derep([x.decode() for x in z["seqs"]], z["abundances"], z["quals"], z["map"])
        except Exception:
            pass
        d = make_sample(err, n, **kw)
        try:
            os.makedirs(cdir, exist_ok=True)
            tmp = path + f".{os.getpid()}.tmp.npz"
            np.savez(tmp, seqs=np.array([x.encode() for x in d.seqs]), abundances=d.abundances, quals=d.quals, map=d.map)
            os.replace(tmp, path)
        except Exception:
            pass
        return d
    return wrapped


def make_inputs(cfg, args, rank, host_inputs=True):
    """Synthetic samples of this rank (SURVEY.md §8d recipe) as HostInput objects + the error matrix."""
    from dada2_amd.api import HostInput
    from dada2_amd.io import extend_err
    from dada2_amd.synth import make_sample, true_variants
    tperr1 = np.load(os.path.join(ROOT, "tests", "golden", "tperr1.npy"))
    c = CONFIGS[cfg]
    n, L, G = args.uniques or c["uniques"], args.length or c["length"], args.variants or c["variants"]
    err = extend_err(tperr1, c.get("q_max", 40))
    kw = dict(L=L, G=G, Lmin=c.get("lmin"), q_hi=c.get("q_hi", 38.0), q_lo=c.get("q_lo", 22.0), q_max=c.get("q_max", 40),
              indel_rate=c.get("indel", 0.0), chunk=200_000 if L <= 500 else 20_000)
    if getattr(args, "deep", False):
        # the recipe of SURVEY.md §8d gives ~1.1 reads per unique (90 % singletons that can never bud); this variant draws the
        # reads at Q34-40, which gives ~14 reads per unique and several times the partitions for the same number of uniques
        kw.update(q_hi=40.0, q_lo=34.0, q_sd=2.0)
    dereps = []
    make_sample = _cached(make_sample)
    if c["samples"] == 1:
        dereps.append(make_sample(err, n, seed=20260925 + cfg + (0 if getattr(args, "shard", False) else 1000 * rank), **kw))
        mine = [0]
    else:
        # configs[3]: 8 samples, half of each sample's true variants shared across samples
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rng = np.random.default_rng(20260925 + cfg)
        shared = true_variants(rng, G // 2, L)
        mine = list(range(rank, c["samples"], world))
        for i in mine:
            own = true_variants(np.random.default_rng(20260925 + cfg + 17 * (i + 1)), G - G // 2, L)
            tv = np.concatenate([shared[0], own[0]])
            tl = np.concatenate([shared[1], own[1]])
            perm = np.random.default_rng(99 + i).permutation(tv.shape[0])   # abundance ranks differ per sample
            dereps.append(make_sample(err, n, seed=20260925 + cfg + 1000 * (i + 1), variants=(tv[perm], tl[perm]), **kw))
    return dereps, ([HostInput.from_derep(d) for d in dereps] if host_inputs else None), err, mine, c


def start_generators(jobs):
    """Draw the sub-records' synthetic samples in child processes (numpy only, one core each) while the headline is measured: they
    land in the input cache, and the sub-records then load them.  jobs: argument lists for `bench.py --gen-only`."""
    import subprocess
    if not os.environ.get("DADA2HIP_BENCH_CACHE", "/tmp/dada2hip_bench_cache"):
        return []
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""          # (they never touch a device)
    return [subprocess.Popen(["nice", "-n", "19", sys.executable, os.path.abspath(__file__), "--gen-only"] + j, env=env,
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for j in jobs]


def sub_config5(api, local, args):
    """BASELINE.json configs[4] inside the default line (so that the driver times it): 200 000 unique ~1.5 kb reads, BAND_SIZE 32."""
    from types import SimpleNamespace
    from dada2_amd.opts import DadaOpts
    a5 = SimpleNamespace(uniques=0, length=0, variants=0, deep=False)
    t0 = time.time()
    dereps, inputs, err, _, c = make_inputs(5, a5, 0)
    gen_s = time.time() - t0
    d = dereps[0]
    o5 = DadaOpts(BAND_SIZE=c["band"])
    api.dada_uniques(inputs[0], None, None, err, None, o5, device=local)   # warm-up
    nst = 2
    t0 = time.perf_counter()
    for _ in range(nst):
        r = api.dada_uniques(inputs[0], None, None, err, None, o5, device=local)
    dt = (time.perf_counter() - t0) / nst
    roof = None
    if not args.no_profile_pass:
        smp = api.Sample(inputs[0], None, None, None, device=local)
        smp.run(err, o5)
        os.environ["DADA2HIP_PROFILE"] = "1"
        pst = smp.run(err, o5).stats
        del os.environ["DADA2HIP_PROFILE"]
        smp.close()
        roof = rooflines(pst, 1510, c["band"], 5, d.nraw)[0]
    st = r.stats
    return {"workload": workload_name(5, d.nraw, max(len(x) for x in d.seqs[:256]), c["band"], c, False), "value": d.nraw / dt, "unit": "uniques/s",
            "ms_per_step": dt * 1e3, "steps": nst, "partitions": r.nclust, "comparisons": st["ncompare"], "comparisons_per_s": st["ncompare"] / dt,
            "nw": st["nnw"], "shrouded_frac": st["nshroud"] / max(1, st["ncompare"]), "roofline": roof, "gen_s": gen_s}


def sub_config4(api, local, args):
    """BASELINE.json configs[3] on ONE GPU inside the default line: 8 samples x 250 000 uniques, two samples in flight."""
    from types import SimpleNamespace
    from dada2_amd.opts import DadaOpts
    a4 = SimpleNamespace(uniques=0, length=0, variants=0, deep=False)
    t0 = time.time()
    dereps, inputs, err, _, c = make_inputs(4, a4, 0)
    gen_s = time.time() - t0
    o4 = DadaOpts(BAND_SIZE=c["band"])
    inflight = max(1, min(args.inflight, len(inputs)))
    api.dada_uniques_multi(inputs, err, o4, devices=(local,) * inflight)   # warm-up
    nst = 2
    t0 = time.perf_counter()
    for _ in range(nst):
        res = api.dada_uniques_multi(inputs, err, o4, devices=(local,) * inflight)
    dt = (time.perf_counter() - t0) / nst
    n = sum(d.nraw for d in dereps)
    return {"workload": workload_name(4, dereps[0].nraw, 250, c["band"], c, False) + f"; all 8 samples on this ONE GPU, {inflight} in flight",
            "value": n / dt, "unit": "uniques/s", "ms_per_step": dt * 1e3, "steps": nst, "samples": len(inputs),
            "partitions_per_sample": [r.nclust for r in res], "comparisons": int(sum(r.stats["ncompare"] for r in res)), "gen_s": gen_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--uniques", type=int, default=0, help="override the config's uniques per sample")
    ap.add_argument("--length", type=int, default=0)
    ap.add_argument("--variants", type=int, default=0)
    ap.add_argument("--band", type=int, default=0)
    ap.add_argument("--selfconsist", action="store_true", help="configs[2]: err from all-ones, noqual refit, resident passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-uniques", type=int, default=0, help="prefix of the sample timed on the CPU (0 = auto, bounded)")
    ap.add_argument("--cpu-full", action="store_true", help="time the reference on the WHOLE sample and check every output against the GPU's")
    ap.add_argument("--no-cpu-whole", action="store_true", help="skip the one reference run on the WHOLE sample (about a minute at 10^6 uniques)")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="timed repetitions of the all-core reference run (best is reported)")
    ap.add_argument("--no-extras", action="store_true", help="skip the selfconsist / secondary_workload sub-records of the default line")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--shard", action="store_true", help="ONE sample, the per-unique work of its uniques split over the ranks "
                    "(dada2hip_sample_run_sharded, DESIGN.md 7): strong scaling of a single dada() call; resident samples")
    ap.add_argument("--inflight", type=int, default=3, help="samples of one rank in flight on its GPU (configs with several samples per rank: "
                    "dada2hip_run_multi with the device listed that many times; their rounds take turns, everything else overlaps)")
    ap.add_argument("--deep", action="store_true", help="workload variant with >= 5 reads per unique (reads drawn at Q34-40)")
    ap.add_argument("--gen-only", action="store_true", help="draw this configuration's synthetic samples into the input cache and exit "
                    "(no GPU touched: the default line starts these for its sub-records while it measures the headline)")
    ap.add_argument("--gen-sample", type=int, default=-1, help="with --gen-only --config 4: only this sample of the eight")
    args = ap.parse_args()
    if args.gen_only:
        if args.gen_sample >= 0:   # one sample of configs[3]'s eight = what rank `gen_sample` of 8 draws
            os.environ["WORLD_SIZE"] = "8"
            make_inputs(args.config, args, args.gen_sample, host_inputs=False)
        else:
            make_inputs(args.config, args, 0, host_inputs=False)
        return

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.shard:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    if world > 1 and "DADA2HIP_HOST_THREADS" not in os.environ:
        # one process per GPU on one host: the marshalling threads of the ranks share the host's cores and memory channels
        os.environ["DADA2HIP_HOST_THREADS"] = str(max(8, min(64, (os.cpu_count() or 64) // world)))
    from dada2_amd import api
    from dada2_amd.opts import DadaOpts

    extras = (args.config == 3 and world == 1 and not args.selfconsist and not args.no_extras and not args.deep and not args.shard
              and not args.uniques)
    gens = []
    t0 = time.time()
    dereps, inputs, err, mine, c = make_inputs(args.config, args, rank)
    t_gen = time.time() - t0
    band = args.band or c["band"]
    opts = DadaOpts(BAND_SIZE=band)
    strong = c["samples"] > 1 or args.shard
    maxcol = err.shape[1]
    shard_smp = None
    if args.shard:   # one resident copy of THE sample per rank; each rank works on its block of the uniques
        from dada2_amd import shard as shardmod
        shard_smp = api.Sample(inputs[0], None, None, None, device=local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ar_ms = []                                    # wall of the all-reduce of each step on this rank (multi-GPU runs explain themselves)

    def allreduce_trans(results):
        local_t = np.zeros((16, maxcol), dtype=np.int64)
        for r in results:
            local_t[:, : r.subqual.shape[1]] += r.subqual
        if world > 1:   # accumulateTrans across samples: the path's only exchange (int64, <= 12 KB)
            ta = time.perf_counter()
            t = torch.from_numpy(local_t).cuda()
            dist.all_reduce(t)
            local_t = t.cpu().numpy()
            ar_ms.append((time.perf_counter() - ta) * 1e3)
        return local_t

    sc_info = None

    def step():
        """One boundary call per sample of this rank (host inputs in, six result objects out)."""
        nonlocal sc_info
        if args.selfconsist:
            tm = []
            res, err_out, errs = api.dada(dereps[0], None, self_consist=True, opts=opts, device=local, timings=tm,
                                          host_input=inputs[0])
            sc_info = {"passes": len(tm) - 1, "ms_create": tm[0], "ms_per_pass": tm[1:], "partitions_last": res.nclust}
            results = [res]
        elif args.shard:
            return [shardmod.dada_sharded(shard_smp, err, opts, dist=dist, collective_device=torch.device("cuda", local))]
        else:
            if len(inputs) > 1 and args.inflight > 1:   # several samples on this rank's GPU: `inflight` boundary calls at a time
                results = api.dada_uniques_multi(inputs, err, opts, devices=(local,) * min(args.inflight, len(inputs)))
            else:
                results = [api.dada_uniques(hi, None, None, err, None, opts, device=local) for hi in inputs]
        allreduce_trans(results)
        return results

    for _ in range(args.warmup):
        step()
    barrier()
    del ar_ms[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        results = step()
    dt_rank = time.perf_counter() - t0            # this rank's own wall, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    n_local = sum(d.nraw for d in dereps)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([n_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(nn)
        total_uniques = n_local if args.shard else int(nn.item())   # --shard: every rank holds the SAME sample
        # per-rank record: own wall before the closing barrier, samples of the rank, time in the all-reduce
        rk = torch.tensor([dt_rank / args.steps * 1e3, float(len(mine)), (sum(ar_ms) / max(1, len(ar_ms))) if ar_ms else 0.0],
                          dtype=torch.float64, device="cuda")
        allrk = [torch.zeros_like(rk) for _ in range(world)]
        dist.all_gather(allrk, rk)
        per_rank = [{"rank": i, "ms_per_step_own_wall": float(v[0]), "samples": int(v[1]), "ms_allreduce_per_step": float(v[2])}
                    for i, v in enumerate(allrk)]
    else:
        total_uniques = n_local
        per_rank = [{"rank": 0, "ms_per_step_own_wall": dt_rank / args.steps * 1e3, "samples": len(mine), "ms_allreduce_per_step": 0.0}]

    if rank == 0:
        res, d = results[0], dereps[0]
        st = res.stats
        value = total_uniques * args.steps / dt
        L = max(len(s) for s in d.seqs[:256])

        # ---- secondary: the same pass on a resident sample + a fully event-timed pass for the roofline --------------
        resident, prof, saturated = None, None, None
        if not args.selfconsist and not args.shard:
            smp = api.Sample(inputs[0], None, None, None, device=local)
            smp.run(err, opts)
            tr0 = time.perf_counter()
            nres = max(1, min(args.steps, 5))
            for _ in range(nres):
                rr = smp.run(err, opts)
            resident = {"ms_per_pass": (time.perf_counter() - tr0) / nres * 1e3, "ms_upload": rr.stats["ms_upload"],
                        "uniques_per_s": d.nraw * nres / (time.perf_counter() - tr0),
                        "note": "sample already resident in HBM (dada2hip_sample_run): what selfConsist passes 2..n cost"}
            if not args.no_profile_pass:
                os.environ["DADA2HIP_PROFILE"] = "1"
                prof = smp.run(err, opts).stats
                del os.environ["DADA2HIP_PROFILE"]
                saturated = nw_saturated(smp, err, opts) if args.config != 5 else None   # (long reads run k_nw_adw)
            smp.close()
        pst = prof or st
        ranked = rooflines(pst, L, band, args.config, d.nraw if prof else 0)
        roofline, other = ranked[0], ranked[1:]
        if prof is None:
            # without the event-timed pass the library only times a sample of the launches: no per-launch figure is claimed
            for r in ranked:
                for k in ("achieved", "frac", "frac_of_measured_peak", "avg_launch_ms", "kernel_ms"):
                    if k in r:
                        r[k] = None
                r["timing"] = "not measured in this run (--no-profile-pass / --selfconsist): see the default run's record"

        # the sub-records' synthetic samples (configs[1] deep, configs[3], configs[4]: 100 s of numpy) are drawn by three niced child
        # processes from HERE on - every GPU figure of the headline has been taken, what follows is the reference's leg on the host
        # cores (it uses 32-64 of them; the box has 256) - and land in the input cache the sub-records load from
        if extras:
            gens = start_generators([["--config", "2", "--deep"], ["--config", "5"], ["--config", "4"]])
        cpu = None
        if not args.no_cpu_baseline and world == 1 and not args.selfconsist and not args.shard:
            cpu = cpu_baseline(d, err, opts, args, res, gpu_cmp_per_s=st["ncompare"] * len(inputs) * world * args.steps / dt)

        # ---- sub-records of the default line: BASELINE configs[2]'s selfConsist loop on this very sample, and a workload
        #      whose comparisons are NOT 98 % shrouded (28 reads per unique) --------------------------------------------
        secondary = plain2 = None
        bimera = None
        sub5 = sub4 = None
        if args.config == 3 and world == 1 and not args.selfconsist and not args.no_extras and not args.deep and not args.shard:
            tm = []
            t_sc = time.perf_counter()
            sc_passes = []

            def sc_pass(k, used, max_clust, results):   # what each pass of the loop was: partitions, calls, pairs, where the host's time went
                ps = results[0].stats
                sc_passes.append({"partitions": int(results[0].nclust), "shuffle_calls": int(ps["nshuffle"]), "stored": int(ps["nstored"]),
                                  "nw": int(ps["nnw"]), "nw_run_for_rounds": int(ps["nnw_run"]), "moves": int(ps["nmoves"]),
                                  "tail_launches": int(ps["tail_launches"]), "batch_compares": int(ps["batch_compares"]),
                                  "ms": {kk[3:]: round(float(ps[kk]), 2) for kk in ("ms_total", "ms_setup", "ms_round0", "ms_bookkeep", "ms_wait_device",
                                                                                   "ms_replay", "ms_enqueue", "ms_final")}})

            res_sc, err_sc, errs_sc = api.dada(dereps[0], None, self_consist=True, opts=opts, device=local, timings=tm,
                                               host_input=inputs[0], on_pass=sc_pass)
            sc_info = {"what": "BASELINE.json configs[2]: learnErrors-style selfConsist loop on the bench sample (err from all-ones with "
                               "MAX_CLUST=1, noqualErrfun refit per pass, resident sample; R/dada.R:256-405)",
                       "passes": len(tm) - 1, "ms_create": tm[0], "ms_per_pass": tm[1:], "ms_total": (time.perf_counter() - t_sc) * 1e3,
                       "per_pass": sc_passes,
                       "partitions_last": res_sc.nclust, "converged": bool(any(np.array_equal(e, err_sc) for e in errs_sc)),
                       "uniques_per_s_whole_loop": d.nraw / (time.perf_counter() - t_sc)}
            for g in gens:
                try:
                    g.wait(timeout=600)
                except Exception:   # noqa: BLE001
                    g.kill()
            secondary = secondary_workload(api, opts, local, args)
            try:
                plain2 = secondary_workload(api, opts, local, args, deep=False)
            except Exception as e:   # noqa: BLE001
                plain2 = {"error": repr(e)}
            bimera = bimera_table_record(api, local, reference=cpu is not None)
            try:
                sub5 = sub_config5(api, local, args)
            except Exception as e:   # noqa: BLE001  (a sub-record never lets the main line down)
                sub5 = {"error": repr(e)}
            try:
                sub4 = sub_config4(api, local, args)
            except Exception as e:   # noqa: BLE001
                sub4 = {"error": repr(e)}

        out = {
            "metric": "unique reads denoised/sec (dada() wall-clock)", "value": value, "unit": "uniques/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "int32 DP + f64 lambda/p-value", "data": "synthetic",
            "timed_region": "dada2hip_dada_uniques boundary call: host inputs -> marshalling, H2D, k-mer build, all rounds, "
                            "final pass, tables, D2H of the six outputs" + ("; selfConsist loop incl. sample upload" if args.selfconsist else ""),
            "config": {"workload": workload_name(args.config, d.nraw, L, band, c, args.selfconsist) +
                                   (" [deep variant: reads drawn at Q34-40, %.1f reads per unique]" % (float(d.abundances.sum()) / d.nraw) if args.deep else ""),
                       "baseline_config": args.config, "uniques_per_sample": d.nraw, "samples_total": c["samples"],
                       "samples_this_rank": len(mine), "reads_per_sample": int(d.abundances.sum()), "partitions": res.nclust,
                       "comparisons": st["ncompare"], "nw": st["nnw"], "gapless": st["ngapless"],
                       "shrouded": st["nshroud"], "greedy_skipped": st["nskipped"], "shuffles": st["nshuffle"],
                       "host_input_bytes": inputs[0].nbytes,
                       "parallelism": ("one sample, its uniques in %d blocks (dada2hip_sample_run_sharded), %s, sample resident"
                                       % (world, "host-driven rounds with the movers / bud candidates exchanged per round" if world > 1
                                          else "world 1: the entry point runs the ordinary device-driven engine") if args.shard
                                       else (f"{c['samples']} samples round-robin over {world} rank(s), {min(args.inflight, len(inputs))} in flight per GPU" if strong else f"sample-per-gpu x{world}")),
                       "shard_collectives_per_step": res.stats.get("shard_collectives") if args.shard else None},
            "roofline": roofline, "roofline_others": other, "roofline_nw_saturated": saturated,
            "roofline_note": "roofline = the kernel class with the largest summed device time of the event-timed pass (phases_ms_last_step."
                             "device_ms_profiled_pass); roofline_others = the remaining classes in that order",
            "cpu_baseline": cpu,
            "resident": resident,
            "selfconsist": sc_info,
            "secondary_workload": secondary,
            "config2_plain_100k": plain2,
            "bimera_table": bimera,
            "config5_long_reads": sub5,
            "config4_eight_samples_one_gpu": sub4,
            "phases_ms_last_step": phases(st, pst if prof else None),
            "comparisons_per_s": st["ncompare"] * len(inputs) * world * args.steps / dt,
            "per_rank": per_rank,
            "samples_in_flight_per_gpu": (min(args.inflight, len(inputs)) if len(inputs) > 1 else 1),
            "collective": "one all-reduce(sum) of the 16 x %d int64 transition counts per step (RCCL; accumulateTrans, R/errorModels.R:462-471)" % maxcol if world > 1 else None,
            "gen_s": t_gen,
        }
        print(json.dumps(out))
    if shard_smp is not None:
        shard_smp.close()
    if dist is not None:
        dist.destroy_process_group()


def workload_name(cfg, n, L, band, c, selfconsist):
    names = {2: "BASELINE.json configs[1]", 3: "BASELINE.json configs[2] size (the headline: 1M uniques x 250 nt)",
             4: "BASELINE.json configs[3]", 5: "BASELINE.json configs[4]"}
    s = f"{c['samples']} x " if c["samples"] > 1 else ""
    return (f"{s}{n} unique {L}-nt synthetic reads ({names[cfg]}), "
            + ("selfConsist loop from an all-ones error matrix with the noqual refit" if selfconsist else "tperr1 fixed error matrix")
            + f", BAND_SIZE {band}, dada() defaults")


def phases(st, prof):
    """Host wall split of the last timed call, and (from the event-timed pass) device time per kernel class."""
    out = {"host_wall_ms": {k: st[k] for k in ("ms_total", "ms_upload", "ms_screen", "ms_bookkeep", "ms_final", "ms_wait_device",
                                                "ms_replay", "ms_enqueue")},
           "moves": st["nmoves"], "batch_compares": st["batch_compares"],
           "chains_without_compare": st["lite_chains"], "of_which_needed_one": st["lite_misses"],
           "alignments": {"committed_nw": st["nnw"], "committed_gapless": st["ngapless"], "run_for_rounds_nw": st["nnw_run"],
                          "run_for_rounds_gapless": st["ngapless_run"], "committed_by_the_rounds_nw": st["nnw_rounds"],
                          "aligned_in_vain_frac": (round(1.0 - st["nnw_rounds"] / st["nnw_run"], 4) if st["nnw_run"] else None),
                          "pointer_free_pass": {"pairs": st["nnw_fast"], "handed_to_the_full_kernel": st["nnw_retry"]},
                          "screen_uniques_through_the_exact_walk": st["screen_stage2"],
                          "note": "committed = the reference's counts (round 0 and the final pass included); run_for_rounds = pairs the aligner "
                                  "processed for the batch compares of the rounds, incl. those a later greedy skip or an unused batch position wasted"},
           "host_wall_note": "upload = marshalling + H2D + k-mer build; screen = enqueue of the compare kernels; "
                             "bookkeep = round tails incl. waiting for the device; final = final pass + outputs"}
    if prof:
        out["device_ms_profiled_pass"] = {k[7:]: prof[k] for k in ("dev_ms_screen", "dev_ms_nw", "dev_ms_shuffle", "dev_ms_pval",
                                                                   "dev_ms_birth", "dev_ms_final", "dev_ms_tail", "dev_ms_pf_screen", "dev_ms_pf_nw")}
        if prof.get("overlap_on"):
            out["overlap"] = {"prefetch_compares": prof["pf_compares"], "centres_prefetched": prof["pf_centres"],
                              "rounds_served_from_a_prefetched_batch": prof["pf_hits"], "waits_inside_the_launch": prof["pf_waits"],
                              "launches_left_for_a_prefetch": prof["pf_exits"], "tail_threads_per_block": prof["tail_threads"],
                              "note": "the next batch's compare runs on a second stream UNDER the persistent tail (pf_screen / pf_nw above "
                                      "lie inside the interval of `tail`): DESIGN.md 5c"}
        out["device_ms_note"] = ("HIP-event time of EVERY launch of a resident pass under DADA2HIP_PROFILE=1, summed per kernel class; tail = the "
                                 "persistent round-tail launches (k3_tail: shuffles, p-update, bud, birth of every round), which replace the "
                                 "shuffle / pval / birth launch chains")
        if prof.get("tail_launches"):
            out["round_tail"] = {"launches": prof["tail_launches"], "blocks_per_launch": prof["tail_blocks"], "pauses": prof["tail_pauses"],
                                 "shuffle_calls": prof["tail_levels"],
                                 "block0_ms": {k[8:]: round(prof[k], 3) for k in ("tail_ms_entry", "tail_ms_shuffle0", "tail_ms_shuffle_more",
                                                                                  "tail_ms_pupdate", "tail_ms_barriers", "tail_ms_birth",
                                                                                  "tail_ms_publish", "tail_ms_release")},
                                 "note": "block 0's wall clock inside the persistent launches: entry = entry barrier, barriers = waiting for the "
                                         "other blocks incl. the serial end of the round (birth) run by the last arriver and block 0's own agent-scope release (release); birth / publish = time "
                                         "the deciding block spent in the serial section / copying the result block to the host"}
    return out


def roofline_tail(st, n_uniques, cfg):
    """The persistent round tail (k3_tail: every b_shuffle2 call, b_p_update, b_bud and the birth of every round inside a few launches;
    /root/reference/src/Rmain.cpp:316-331 behind the compare).  Its ALGORITHMIC bytes are what its scans must read - 15 B per unique in
    a round's commit + first shuffle call, 16 B in every later call, 12 B in the p-update - against the HBM peak; the
    model it is actually bound by is printed beside it: a round is a chain of dependent global round trips and grid barriers, so
    `us_per_round` and `dependent_phases_per_round` are the figures to watch, not the bandwidth."""
    ms = st.get("dev_ms_tail", 0.0)
    if not ms or not st.get("tail_launches"):
        return None
    rounds = max(1, int(st["rounds"]) - 1)                      # rounds behind round 0 (each: commit + shuffles + p-update + bud)
    calls = int(st["nshuffle"])                                 # b_shuffle2 calls of the run (the first of each round is the commit)
    later = max(0, calls - rounds)
    evals = rounds + 1                                          # (+ the evaluation behind round 0)
    alg = float(n_uniques) * (15.0 * rounds + 16.0 * later + 12.0 * evals)
    nl = max(1, int(st["tail_launches"]))
    phases = calls + evals                                      # grid-wide phases, each closed by a grid barrier
    r = {"kernel": "k3_tail (persistent round tail: b_shuffle2 x n, b_p_update, b_bud, birth of every round)", "bound": "hbm",
         "achieved": alg / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "traffic": None,
         "kernel_ms": ms, "launches": int(st["tail_launches"]), "avg_launch_ms": ms / nl, "algorithmic_bytes_per_launch": alg / nl,
         "timing": "every launch event-timed (DADA2HIP_PROFILE=1)",
         "latency_model": {"rounds": rounds, "shuffle_calls": calls, "grid_phases": phases, "us_per_round": ms * 1e3 / rounds,
                           "us_per_phase": ms * 1e3 / max(1, phases), "blocks": int(st.get("tail_blocks", 0)),
                           "threads_per_block": int(st.get("tail_threads", 0) or 1024),
                           "note": "a phase = one scan over the uniques + a work-list pass + a grid barrier; each is a chain of ~4-10 dependent "
                                   "global round trips of ~2 us (block 0's clock per phase: round_tail.block0_ms)"}}
    r["frac"] = r["achieved"] / r["peak"]
    peaks = measured_peaks()
    if peaks and "hbm_read_gbs" in peaks:
        r["peak_measured"] = peaks["hbm_read_gbs"]
        r["frac_of_measured_peak"] = r["achieved"] / peaks["hbm_read_gbs"]
    tr = load_traffic(cfg)
    if tr and tr.get("tail"):
        r["traffic"] = tr["tail"].get("hbm_bytes_per_launch")
        r["traffic_source"] = "committed rocprofv3 PMC pass of this command: " + tr["_file"]
        if r["traffic"]:
            # the scans are served from L2 / MALL (and, since round 6, mostly from the tail's LDS mirror): what binds the kernel is
            # its chain of dependent round trips and grid barriers (latency_model), not the HBM rate `frac` is quoted against
            r["hbm_is_not_the_bound"] = {"traffic_over_algorithmic": round(r["traffic"] / r["algorithmic_bytes_per_launch"], 3),
                                         "bound_by": "dependent round trips + grid barriers: latency_model.us_per_round"}
    return r


def rooflines(st, L, band, cfg, n_uniques=0):
    """Per-launch rooflines of the hot kernel classes from HIP-event times (exact sums under DADA2HIP_PROFILE=1), the one with the
    largest summed device time FIRST - the persistent round tail included (it is half of a pass)."""
    peaks = measured_peaks()
    nw_ms, nw_n = st["nw_kernel_ms"], max(1, st["nw_kernel_launches"])
    sc_ms, sc_n = st["screen_kernel_ms"], max(1, st["screen_kernel_launches"])
    nw_ops = st["nw_cells"] * INT_OPS_PER_CELL
    # algorithmic bytes the screen must read per compared unique: its ordered k-mer record row 2*(L-4) B + 6 B of
    # scalars (the reference streams 1 544 B for the same decision) - none for greedy-skipped uniques
    screen_bytes = st["screen_bytes"]      # counted by the library: bytes its screen launches had to read (see dada2hip.h)
    sampled = bool(st.get("kernel_times_sampled", 1))
    timing = "extrapolated from sampled launches" if sampled else "every launch event-timed (DADA2HIP_PROFILE=1)"
    roof_nw = {"kernel": "k_nw_ad / k_nw_adw (banded NW + traceback + lambda)", "bound": "valu",
               "achieved": nw_ops / (nw_ms * 1e-3) / 1e12 if nw_ms > 0 else 0.0, "peak": PEAK_VALU_TOPS, "unit": "Tops/s",
               "traffic": None, "avg_launch_ms": nw_ms / nw_n, "launches": st["nw_kernel_launches"], "kernel_ms": nw_ms,
               "timing": timing, "algorithmic_ops_per_launch": nw_ops / nw_n}
    roof_sc = {"kernel": "k2_screen_multi (k-mer screen of every unique against <= 8 centres per pass; round 0: k_screen)", "bound": "hbm",
               "achieved": screen_bytes / (sc_ms * 1e-3) / 1e9 if sc_ms > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "traffic": None, "avg_launch_ms": sc_ms / sc_n, "launches": st["screen_kernel_launches"], "kernel_ms": sc_ms,
               "timing": timing, "algorithmic_bytes_per_launch": screen_bytes / sc_n}
    for r, key in ((roof_nw, "valu_int32_tops"), (roof_sc, "hbm_read_gbs")):
        r["frac"] = r["achieved"] / r["peak"]
        if peaks and key in peaks:
            r["peak_measured"] = peaks[key]
            r["frac_of_measured_peak"] = r["achieved"] / peaks[key]
            r["peak_measured_source"] = peaks["_file"]
    tr = load_traffic(cfg)
    if tr:
        roof_nw["traffic"] = tr.get("nw", {}).get("hbm_bytes_per_launch")
        roof_sc["traffic"] = tr.get("screen", {}).get("hbm_bytes_per_launch")
        roof_nw["traffic_source"] = roof_sc["traffic_source"] = "committed rocprofv3 PMC pass of this command: " + tr["_file"]
    ranked = [roof_nw, roof_sc]
    rt = roofline_tail(st, n_uniques, cfg) if n_uniques else None
    if rt:
        ranked.append(rt)
    ranked.sort(key=lambda r: -(r.get("kernel_ms") or 0.0))
    return ranked


def nw_saturated(smp, err, opts, target=60000, reps=3):
    """The NW kernel of the rounds (k_nw_ad) on ONE saturating launch: a b_compare round of the resident sample against
    its first centre (dada2hip_sample_compare, event-timed launch), the alignment batch thinned with the skip mask to
    just under the 65 536 the entry gives to this kernel.  The per-round launches of the timed pass hold ~9 k alignments
    (under three waves per SIMD): this is the same kernel where latency is not the excuse."""
    old = os.environ.get("DADA2HIP_NW_KERNEL")
    os.environ["DADA2HIP_NW_KERNEL"] = "coop"
    try:
        _, _, cls, _ = smp.compare(0, err, opts)
        idx = np.flatnonzero(cls == 3)
        if idx.size == 0:
            return None
        keep = idx if idx.size <= target else np.random.default_rng(0).choice(idx, size=target, replace=False)
        skip = np.ones(cls.size, dtype=np.uint8)
        skip[keep] = 0
        best = None
        for _ in range(reps):
            _, _, _, st = smp.compare(0, err, opts, skip=skip)
            if best is None or st["nw_kernel_ms"] < best["nw_kernel_ms"]:
                best = st
    finally:
        if old is None:
            del os.environ["DADA2HIP_NW_KERNEL"]
        else:
            os.environ["DADA2HIP_NW_KERNEL"] = old
    ops = best["nw_cells"] * INT_OPS_PER_CELL
    r = {"kernel": "k_nw_ad, one launch", "alignments": int(best["nnw"]), "launch_ms": best["nw_kernel_ms"], "bound": "valu",
         "achieved": ops / (best["nw_kernel_ms"] * 1e-3) / 1e12, "peak": PEAK_VALU_TOPS, "unit": "Tops/s"}
    r["frac"] = r["achieved"] / r["peak"]
    peaks = measured_peaks()
    if peaks and "valu_int32_tops" in peaks:
        r["peak_measured"] = peaks["valu_int32_tops"]
        r["frac_of_measured_peak"] = r["achieved"] / peaks["valu_int32_tops"]
    return r


def load_traffic(cfg):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_traffic_cfgN.json) of this same command;
    bench.py cannot run the profiler on itself.  None when no profile of this config is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_traffic_cfg{cfg}.json")))
    if not files:
        return None
    t = json.load(open(files[-1]))
    t["_file"] = os.path.relpath(files[-1], ROOT)
    return t


def secondary_workload(api, opts, local, args, deep=True):
    """A second workload in the same line: BASELINE configs[1]'s 100 k uniques - plain (deep=False: the survey recipe, 1.1 reads
    per unique), or drawn at Q34-40 (28 reads per unique), where a third of the comparisons survive the k-mer screen instead of 1.7 %."""
    from types import SimpleNamespace
    a2 = SimpleNamespace(uniques=0, length=0, variants=0, deep=deep)
    t0 = time.time()
    dereps, inputs, err, _, c = make_inputs(2, a2, 0)
    gen_s = time.time() - t0
    d = dereps[0]
    r = api.dada_uniques(inputs[0], None, None, err, None, opts, device=local)   # warm-up
    nst = 3
    t0 = time.perf_counter()
    for _ in range(nst):
        r = api.dada_uniques(inputs[0], None, None, err, None, opts, device=local)
    dt = (time.perf_counter() - t0) / nst
    st = r.stats
    smp = api.Sample(inputs[0], None, None, None, device=local)
    smp.run(err, opts)
    t0 = time.perf_counter()
    smp.run(err, opts)
    t_res = time.perf_counter() - t0
    roof = None
    if not args.no_profile_pass:
        os.environ["DADA2HIP_PROFILE"] = "1"
        pst = smp.run(err, opts).stats
        del os.environ["DADA2HIP_PROFILE"]
        roof = [r for r in rooflines(pst, 250, opts.BAND_SIZE, 2) if r["bound"] == "valu"][0]
        roof["traffic"] = None
        roof.pop("traffic_source", None)
    smp.close()
    ncmp = st["ncompare"]
    return {"workload": "%d unique 250-nt synthetic reads (BASELINE.json configs[1]%s: %.1f reads per unique, "
                        "tperr1 fixed error matrix, BAND_SIZE %d" % (d.nraw, " size) drawn at Q34-40" if deep else ", the survey's recipe)", float(d.abundances.sum()) / d.nraw, opts.BAND_SIZE),
            "value": d.nraw / dt, "unit": "uniques/s", "ms_per_step": dt * 1e3, "steps": nst, "ms_resident_pass": t_res * 1e3,
            "partitions": r.nclust, "comparisons": ncmp, "comparisons_per_s": ncmp / dt,
            "shrouded_frac": st["nshroud"] / max(1, ncmp), "nw": st["nnw"], "gapless": st["ngapless"],
            "greedy_skipped": st["nskipped"],
            "aligned_in_vain_frac": (round(1.0 - st["nnw_rounds"] / st["nnw_run"], 4) if st.get("nnw_run") else None),
            "roofline_nw": roof, "gen_s": gen_s}


def bimera_table_record(api, device, reference=True):
    """The row after the path (removeBimeraDenovo's C_table_bimera2): 3 000 sequences x 8 samples, one call.
    With `reference` (part of the cpu_baseline leg: off under --no-cpu-baseline) the reference on all host cores beside it, tables
    compared.  Never lets the main line down."""
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import bench_bimera
        return bench_bimera.record(api, device=device, reference=reference)
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)}


def cpu_baseline(d, err, opts, args, gpu_res, gpu_cmp_per_s=None):
    """Reference C++ (oracle/_ref; falls back to the C restatement, kind 'port') on this box's host cores."""
    from oracle import ref, cport
    ncores = os.cpu_count() or 1
    L = max(len(x) for x in d.seqs[:64])
    if args.cpu_full:
        n = d.nraw
    elif args.cpu_uniques:
        n = min(d.nraw, args.cpu_uniques)
    else:   # ~10-20 s of all-core CPU work: N x C grows roughly as N^1.5 on these samples
        n = min(d.nraw, 200_000 if L <= 500 else 6_000)
    seqs, ab, q = d.seqs[:n], d.abundances[:n], d.quals[:n]
    out = {"unit": "uniques/s", "cores": ncores,
           "sample": ("the whole bench sample" if n == d.nraw else f"first {n} uniques of the bench sample (abundance-sorted prefix)")
                     + ", full dada_uniques, multithread=TRUE"}
    if not ref.available():
        out["kind"] = "port"
        out["cores"] = 1
        n1 = min(n, 20_000)
        t0 = time.perf_counter()
        cport.dada_uniques(seqs[:n1], ab[:n1], None, err, q[:n1], opts)
        t1 = time.perf_counter() - t0
        out.update(value=n1 / t1, seconds=t1, sample=f"first {n1} uniques of the bench sample, full dada_uniques, scalar C port")
        return out
    out["kind"] = "reference"
    out["threading"] = ("the reference's own parallelFor call sites (b_compare_parallel, FinalSubsParallel) on a persistent worker "
                        "pool (oracle/shim/RcppParallel.h + ref_capi.cpp), the stand-in for RcppParallel's TBB pool")

    packed = ref.pack_inputs(seqs, ab, None, err, q)     # marshalling outside the timed call, as for the GPU side

    def timed(flavour, nthreads, reps, sub=None):
        best, r = None, None
        for _ in range(max(1, reps)):
            ref.set_threads(nthreads)
            cs = []
            if sub is None:
                r = ref.dada_uniques(seqs, ab, None, err, q, opts, multithread=True, flavour=flavour, packed=packed, call_seconds=cs)
            else:
                r = ref.dada_uniques(seqs[:sub], ab[:sub], None, err, q[:sub], opts, multithread=True, flavour=flavour, call_seconds=cs)
            best = cs[-1] if best is None else min(best, cs[-1])
        return best, r

    # thread-count sweep on a quarter of the sample (the memory-bound shuffles and the serial b_bud do not scale with cores)
    cand = sorted({t for t in (32, 64, 128, 256, ncores) if t <= ncores} or {ncores})
    nsw = n if n <= 20_000 else max(20_000, n // 4)
    sweep = {}
    for t in cand:
        ts, _ = timed("O2", t, 1, sub=None if nsw == n else nsw)
        sweep[str(t)] = nsw / ts
    tbest = int(max(sweep, key=lambda k: sweep[k]))
    out["sweep"] = {"uniques": nsw, "uniques_per_s_by_threads": sweep}
    out["threads_best"] = tbest
    out["cores"] = tbest                 # the threads the timed run used (the box has `host_cores`)
    out["host_cores"] = ncores

    t_all, r = timed("O2", tbest, args.cpu_repeats)
    out.update(value=n / t_all, seconds=t_all, partitions=r.nclust, build="-O2 (R's default flags)", repeats=max(1, args.cpu_repeats),
               comparisons_per_s=n * r.nclust / t_all)
    if gpu_cmp_per_s:
        out["gpu_comparisons_per_s"] = gpu_cmp_per_s
        out["comparisons_per_s_ratio_gpu_over_cpu"] = gpu_cmp_per_s / out["comparisons_per_s"]
    if ref.available("O3"):
        t3, r3 = timed("O3", tbest, 1)
        out["O3"] = {"value": n / t3, "seconds": t3, "build": "-O3 -march=x86-64-v3", "comparisons_per_s": n * r3.nclust / t3}
    # single thread on a smaller prefix so the default run stays within minutes
    n1 = min(n, max(500, int(20_000 * (250.0 / max(L, 250)) ** 2)))   # ~5-20 s of scalar CPU work at any read length
    ref.set_threads(1)
    cs = []
    r1 = ref.dada_uniques(seqs[:n1], ab[:n1], None, err, q[:n1], opts, multithread=False, call_seconds=cs)
    t1 = cs[-1]
    out["single_thread"] = {"value": n1 / t1, "seconds": t1, "sample_uniques": n1, "partitions": r1.nclust,
                            "comparisons_per_s": n1 * r1.nclust / t1}
    # ... and ONCE the whole sample (no extrapolated baseline): one run at the best thread count of the sweep, every
    # output compared with the GPU's.  About a minute at 10^6 uniques x 250 nt; long reads keep the prefix (--cpu-full there).
    if n < d.nraw and L <= 500 and not getattr(args, "no_cpu_whole", False):
        ref.set_threads(tbest)
        cs = []
        rw = ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, opts, multithread=True, call_seconds=cs)
        whole = {"uniques": d.nraw, "value": d.nraw / cs[-1], "seconds": cs[-1], "threads": tbest, "partitions": rw.nclust,
                 "comparisons_per_s": d.nraw * rw.nclust / cs[-1], "repeats": 1, "build": "-O2 (R's default flags)"}
        if gpu_cmp_per_s:
            whole["comparisons_per_s_ratio_gpu_over_cpu"] = gpu_cmp_per_s / whole["comparisons_per_s"]
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import assert_results_equal
        try:
            assert_results_equal(gpu_res, rw)
            whole["parity_vs_gpu"] = True
        except AssertionError as e:
            whole["parity_vs_gpu"] = False
            whole["parity_error"] = str(e)[:300]
        # ... and THAT is the baseline's headline: the unextrapolated whole-sample run on top, the prefix timings
        # (sweep, repeats, -O3, one thread) beside it
        out["prefix"] = {k: out[k] for k in ("value", "seconds", "partitions", "build", "repeats", "comparisons_per_s", "sample") if k in out}
        out["prefix"]["comparisons_per_s_ratio_gpu_over_cpu"] = out.get("comparisons_per_s_ratio_gpu_over_cpu")
        out.update(value=whole["value"], seconds=whole["seconds"], partitions=whole["partitions"], repeats=1,
                   comparisons_per_s=whole["comparisons_per_s"], cores=tbest,
                   sample="the WHOLE bench sample (%d uniques), one run of the reference's dada_uniques, multithread=TRUE at the best thread "
                          "count of the sweep; every output compared with the GPU's (parity_vs_gpu)" % d.nraw)
        if gpu_cmp_per_s:
            out["comparisons_per_s_ratio_gpu_over_cpu"] = whole["comparisons_per_s_ratio_gpu_over_cpu"]
        out["parity_vs_gpu"] = whole["parity_vs_gpu"]
        out["whole_sample"] = whole
    if n == d.nraw:   # same input as the GPU run: EVERY output compared (tests/helpers.assert_results_equal)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import assert_results_equal
        try:
            assert_results_equal(gpu_res, r)
            out["parity_vs_gpu"] = True
            out["parity_check"] = "full: all six outputs (bit-exact integers/strings, p-values within 1e-10)"
        except AssertionError as e:
            out["parity_vs_gpu"] = False
            out["parity_error"] = str(e)[:300]
    return out


if __name__ == "__main__":
    main()
