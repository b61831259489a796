#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: unique reads denoised / s of dada() wall-clock.

One "step" = one full dada_uniques pass (all divisive rounds + final alignments + output
tables) over one synthetic sample whose reads, qualities and k-mer records are already
resident in HBM (dada2_amd.api.Sample) when the timed region starts.  Workload at N=1 is
BASELINE.json configs[1]: 100 k unique 250-nt synthetic reads, fixed error matrix (tperr1).
With --gpus N every rank denoises its own sample of the same size (weak scaling: the path
shards at sample granularity) and the only collective is the RCCL all-reduce of
the 16 x Q transition-count matrix (accumulateTrans, R/errorModels.R:462-471), inside the
timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel by HIP-event time, algorithmic work / measured duration vs peak
  cpu_baseline  the reference's own C++ (oracle/_ref, multithread=TRUE on all host cores, and
                1 thread) timed on the same box on the same sample (bounded)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_VALU_TOPS = 39.3       # 256 CU x 4 SIMD x 16... = 256 x 64 lanes x 2.4 GHz 32-bit lane-ops/s
INT_OPS_PER_CELL = 8        # fixed algorithmic constant, SURVEY.md §8d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--uniques", type=int, default=100_000)
    ap.add_argument("--length", type=int, default=250)
    ap.add_argument("--variants", type=int, default=256)
    ap.add_argument("--band", type=int, default=16)
    ap.add_argument("--lmin", type=int, default=0, help="ragged true-variant lengths in [lmin, length] (long-read configs)")
    ap.add_argument("--q-hi", type=float, default=38.0)
    ap.add_argument("--q-lo", type=float, default=22.0)
    ap.add_argument("--q-sd", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-uniques", type=int, default=0, help="prefix of the sample timed on the CPU (0 = auto)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from dada2_amd import api
    from dada2_amd.opts import DadaOpts
    from dada2_amd.synth import make_sample

    tperr1 = np.load(os.path.join(ROOT, "tests", "golden", "tperr1.npy"))
    opts = DadaOpts(BAND_SIZE=args.band)
    t0 = time.time()
    d = make_sample(tperr1, args.uniques, L=args.length, G=args.variants, seed=20260925 + 2 + 1000 * rank,
                    Lmin=args.lmin or None, q_hi=args.q_hi, q_lo=args.q_lo, q_sd=args.q_sd,
                    chunk=200_000 if args.length <= 500 else 20_000)
    t_gen = time.time() - t0
    smp = api.Sample.from_derep(d, device=local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        r = smp.run(tperr1, opts)
        if world > 1:   # accumulateTrans across samples: the path's only exchange (int64, <= 12 KB)
            t = torch.from_numpy(r.subqual.astype(np.int64)).cuda()
            dist.all_reduce(t)
            t.cpu()
        return r

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([d.nraw], dtype=torch.int64, device="cuda")
        dist.all_reduce(nn)
        total_uniques = int(nn.item())
    else:
        total_uniques = d.nraw

    if rank == 0:
        st = res.stats
        value = total_uniques * args.steps / dt
        # ---- roofline of the dominant kernel (per launch, from the HIP-event sums of the last step) ----
        nw_ms, nw_n = st["nw_kernel_ms"], max(1, st["nw_kernel_launches"])
        sc_ms, sc_n = st["screen_kernel_ms"], max(1, st["screen_kernel_launches"])
        L = args.length
        nw_ops = st["nw_cells"] * INT_OPS_PER_CELL
        screen_bytes = st["ncompare"] * (2 * (L - 4) + 6) - st["nskipped"] * 2 * (L - 4)
        roof_nw = {"kernel": "k_nw", "bound": "valu", "achieved": nw_ops / (nw_ms * 1e-3) / 1e12 if nw_ms > 0 else 0.0,
                   "peak": PEAK_VALU_TOPS, "unit": "Tops/s", "traffic": None,
                   "avg_launch_ms": nw_ms / nw_n, "launches": st["nw_kernel_launches"]}
        roof_nw["frac"] = roof_nw["achieved"] / roof_nw["peak"]
        roof_sc = {"kernel": "k_screen", "bound": "hbm", "achieved": screen_bytes / (sc_ms * 1e-3) / 1e9 if sc_ms > 0 else 0.0,
                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "traffic": None,
                   "avg_launch_ms": sc_ms / sc_n, "launches": st["screen_kernel_launches"]}
        roof_sc["frac"] = roof_sc["achieved"] / roof_sc["peak"]
        # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command (profiles/*_traffic.json;
        # bench.py cannot run the profiler on itself) — null when no matching profile is committed
        tr = load_traffic()
        if tr and args.uniques == 100_000 and args.length == 250:
            coop = [v for k, v in tr.items() if k.startswith("void k_nw_ad") and isinstance(v, dict)]   # the per-round NW kernel
            roof_nw["traffic"] = max(coop, key=lambda v: v.get("dispatches", 0)).get("hbm_bytes_per_launch") if coop else None
            roof_sc["traffic"] = tr.get("k_screen", {}).get("hbm_bytes_per_launch")
            roof_nw["traffic_source"] = roof_sc["traffic_source"] = tr.get("_file")
        roofline = roof_nw if nw_ms >= sc_ms else roof_sc
        other = roof_sc if nw_ms >= sc_ms else roof_nw

        # ---- CPU baseline: the reference itself on this box's host cores, same sample (bounded) ----
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(d, tperr1, opts, args.cpu_uniques, res)

        out = {
            "metric": "unique reads denoised/sec (dada() wall-clock)", "value": value, "unit": "uniques/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 DP + f64 lambda/p-value",
            "data": "synthetic",
            "config": {"workload": f"{args.uniques} unique {L}-nt synthetic reads per GPU "
                                   f"({'BASELINE.json configs[1] recipe' if L <= 500 else 'long-read shape of BASELINE.json configs[4]'}), "
                                   f"tperr1 fixed error matrix, BAND_SIZE {args.band}, dada() defaults",
                       "uniques_per_gpu": d.nraw, "reads_per_gpu": int(d.abundances.sum()), "partitions": res.nclust,
                       "comparisons": st["ncompare"], "nw": st["nnw"], "gapless": st["ngapless"],
                       "shrouded": st["nshroud"], "greedy_skipped": st["nskipped"], "parallelism": f"sample-per-gpu x{world}"},
            "roofline": roofline, "roofline_secondary": other,
            "cpu_baseline": cpu,
            "phases_ms_last_step": {k: st[k] for k in ("ms_total", "ms_screen", "ms_nw", "ms_bookkeep", "ms_pval", "ms_final")},
            "comparisons_per_s": st["ncompare"] * world * args.steps / dt if world == 1 else None,
            "gen_s": t_gen,
        }
        print(json.dumps(out))
    smp.close()
    if world > 1:
        dist.destroy_process_group()


def load_traffic():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    t = json.load(open(files[-1]))
    k = dict(t.get("kernels", {}))
    k["_file"] = os.path.relpath(files[-1], ROOT)
    return k


def cpu_baseline(d, err, opts, cpu_uniques, gpu_res):
    """Reference C++ (oracle/_ref; falls back to the C restatement, kind 'port') on the host cores."""
    from oracle import ref, cport
    ncores = os.cpu_count() or 1
    n = cpu_uniques or min(d.nraw, 100_000)
    seqs, ab, q = d.seqs[:n], d.abundances[:n], d.quals[:n]
    out = {"unit": "uniques/s", "cores": ncores,
           "sample": f"first {n} uniques of the bench sample (abundance-sorted prefix), full dada_uniques"}
    if ref.available():
        out["kind"] = "reference"
        ref.set_threads(ncores)
        t0 = time.perf_counter()
        r = ref.dada_uniques(seqs, ab, None, err, q, opts, multithread=True)
        t_all = time.perf_counter() - t0
        out["value"] = n / t_all
        out["seconds"] = t_all
        out["partitions"] = r.nclust
        # single thread on a smaller prefix so the default run stays within minutes
        L = max(len(x) for x in seqs[:64])
        n1 = min(n, max(500, int(20_000 * (250.0 / max(L, 250)) ** 2)))   # ~10-30 s of scalar CPU work at any read length
        ref.set_threads(1)
        t0 = time.perf_counter()
        r1 = ref.dada_uniques(seqs[:n1], ab[:n1], None, err, q[:n1], opts, multithread=False)
        t1 = time.perf_counter() - t0
        out["single_thread"] = {"value": n1 / t1, "seconds": t1, "sample_uniques": n1, "partitions": r1.nclust}
        if n == d.nraw:   # same input as the GPU run: parity of the headline outputs, for the record
            out["parity_vs_gpu"] = bool(r.clustering["sequence"] == gpu_res.clustering["sequence"]
                                        and np.array_equal(r.map, gpu_res.map))
    else:
        out["kind"] = "port"
        out["cores"] = 1
        n1 = min(n, 20_000)
        t0 = time.perf_counter()
        r1 = cport.dada_uniques(seqs[:n1], ab[:n1], None, err, q[:n1], opts)
        t1 = time.perf_counter() - t0
        out["value"] = n1 / t1
        out["seconds"] = t1
        out["sample"] = f"first {n1} uniques of the bench sample, full dada_uniques, scalar C port"
    return out


if __name__ == "__main__":
    main()
