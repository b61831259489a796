#!/usr/bin/env python3
"""Condense rocprofv3 (rocpd sqlite) outputs into profiles/<tag>_summary.md + <tag>_kernel_stats.csv.
usage: summarize.py <dir with trace/ pmc_fetch/ pmc_write/ pmc_valu/> <tag> [outdir] [description of the profiled command]"""
import csv, glob, os, sqlite3, sys

out, tag = sys.argv[1], sys.argv[2]
root = sys.argv[3] if len(sys.argv) > 3 else os.path.dirname(os.path.abspath(__file__))
desc = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass"
os.makedirs(root, exist_ok=True)
L = [f"# rocprofv3 summary `{tag}`", "",
     f"Command profiled: `{desc}` (one rocprofv3 run per section, PMC passes separate from the trace).", ""]


def short(n):
    n = n.replace("d2::", "")
    return n.split("(")[0][:60]


dbs = glob.glob(os.path.join(out, "trace", "*.db"))
if dbs:
    db = sqlite3.connect(dbs[0])
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        w.writerows(rows)
    L += ["## Kernel stats (`rocprofv3 --kernel-trace --stats`)", "", "| kernel | calls | total ms | avg us | % of GPU time |", "|---|---|---|---|---|"]
    for n, c, tot, avg, pct in rows:
        L.append(f"| `{short(n)}` | {c} | {tot / 1e3:.3f} | {avg:.2f} | {pct:.2f} |")
    L.append("")
    if any("k2_pf_gate" in n for n, *_ in rows):
        L += ["(`k2_pf_gate` is ONE lane that waits on the second stream for the persistent tail's next plan - the first kernel of a "
              "prefetch compare's chain, enqueued one plan ahead, DESIGN.md 5c: its duration is waiting, not work, and it runs beside "
              "`k3_tail`, whose own durations include its barrier waits.  The percentages are of the summed kernel time of two streams, "
              "not of wall time.)", ""]
    # The per-round screen / NW kernels are also launched speculatively behind k_auto_birth, before the host has seen the
    # bud decision; when no birth was applied on the device those launches find nothing to do and return in a few us.
    # bench.py's HIP-event averages exclude them, so the comparable rocprofv3 average is over the working launches.
    L += ["## Hot kernels without the no-op speculative launches (duration >= 8 us)", "",
          "| kernel | working launches | avg us | no-op launches | avg us |", "|---|---|---|---|---|"]
    for pat in ("%k_nw_ad<%", "%k_nw_adw<%", "%k_screen(%", "%k2_screen_multi%", "%k2_shuffle<%"):
        for n, c1, a1, c0, a0 in db.execute(
                "select name, sum(d >= 8000), avg(case when d >= 8000 then d end), sum(d < 8000), avg(case when d < 8000 then d end) "
                "from (select name, (end - start) as d from kernels where name like ?) group by name", (pat,)):
            L.append(f"| `{short(n)}` | {c1} | {(a1 or 0) / 1e3:.2f} | {c0} | {(a0 or 0) / 1e3:.2f} |")
    L.append("")
    # distribution of the per-launch durations and the idle gap the GPU shows AFTER each kernel (next start - this end):
    # the round chain is a string of dependent launches, so the gaps are part of a round's latency
    ks = list(db.execute("select name, start, end from kernels order by start"))
    import collections
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    for i, (n, st, en) in enumerate(ks):
        dur[short(n)].append((en - st) / 1e3)
        if i + 1 < len(ks):
            g = (ks[i + 1][1] - en) / 1e3
            if g < 200.0:
                gap[short(n)].append(g)
    edges = [0, 4, 6, 8, 10, 12, 15, 20, 25, 30, 40, 60, 100, 200, 10 ** 9]
    L += ["## Per-launch duration histogram (us) and idle gap behind each kernel", "",
          "| kernel | launches | " + " | ".join(f"<{e}" if e < 10 ** 9 else ">=200" for e in edges[1:]) + " | median us | avg gap after (us) | gaps total ms |", "|---|---|" + "---|" * (len(edges) + 2)]
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if len(v) < 50:
            continue
        h = [0] * (len(edges) - 1)
        for x in v:
            for b in range(len(edges) - 1):
                if edges[b] <= x < edges[b + 1]:
                    h[b] += 1
                    break
        sv = sorted(v)
        g = gap.get(n, [])
        L.append(f"| `{n}` | {len(v)} | " + " | ".join(str(x) for x in h) + f" | {sv[len(sv) // 2]:.1f} | {(sum(g) / len(g)) if g else 0:.2f} | {sum(g) / 1e3:.2f} |")
    L.append("")
    tot_busy = sum(sum(v) for v in dur.values()) / 1e3
    tot_gap = sum(sum(v) for v in gap.values()) / 1e3
    L += [f"GPU busy {tot_busy:.1f} ms, idle gaps between consecutive kernels (< 200 us each) {tot_gap:.1f} ms.", ""]
    # the long gaps (200 us .. 20 ms: the device waits for the host inside a pass; longer ones are between passes / steps)
    big = collections.defaultdict(list)
    for i, (n, st, en) in enumerate(ks[:-1]):
        g = (ks[i + 1][1] - en) / 1e3
        if 200.0 <= g < 20000.0:
            big[(short(n), short(ks[i + 1][0]))].append(g)
    if big:
        L += ["## Long idle gaps (200 us - 20 ms): the device waiting for the host", "", "| kernel before | kernel after | gaps | total ms | median us |", "|---|---|---|---|---|"]
        for (a, b2), v in sorted(big.items(), key=lambda kv: -sum(kv[1]))[:12]:
            sv = sorted(v)
            L.append(f"| `{a}` | `{b2}` | {len(v)} | {sum(v) / 1e3:.2f} | {sv[len(sv) // 2]:.0f} |")
        L += ["", f"Total {sum(sum(v) for v in big.values()) / 1e3:.1f} ms in {sum(len(v) for v in big.values())} gaps.", ""]
    # Concurrency: which kernels ran INSIDE the interval of a persistent round-tail launch (the prefetch compares of the second
    # stream)?  For every working k3_tail launch: the kernels whose [start, end] intersects it, by name.
    tails = [(st, en) for n, st, en in ks if "k3_tail" in n and en - st >= 20000]
    if tails:
        import bisect
        starts = [t[0] for t in tails]
        inside = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for n, st, en in ks:
            if "k3_tail" in n:
                continue
            i = bisect.bisect_right(starts, en) - 1
            while i >= 0 and tails[i][1] > st:
                ov = min(en, tails[i][1]) - max(st, tails[i][0])
                if ov > 0:
                    rec = inside[short(n)]
                    rec[0] += 1; rec[1] += ov / 1e3; rec[2] += (en - st) / 1e3
                i -= 1
        tail_ms = sum(en - st for st, en in tails) / 1e6
        L += ["## Kernels that ran inside the interval of a `k3_tail` launch (two streams: the next batch's compare under the round tail)", "",
              f"{len(tails)} working `k3_tail` launches, {tail_ms:.1f} ms in all.", "",
              "| kernel | launches overlapping a tail launch | overlapped ms | their total ms |", "|---|---|---|---|"]
        for n, (c, ov, tot) in sorted(inside.items(), key=lambda kv: -kv[1][1]):
            L.append(f"| `{n}` | {c} | {ov / 1e3:.2f} | {tot / 1e3:.2f} |")
        L += ["", f"Overlapped kernel time {sum(v[1] for v in inside.values()) / 1e3:.1f} ms of {tail_ms:.1f} ms of tail launches.", ""]
    res = list(db.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                          "max(workgroup_x), avg(grid_x) from kernels where name like 'd2::%' or name like 'void d2::%' group by name"))
    L += ["## Dispatch resources", "", "| kernel | VGPR | AGPR | SGPR | LDS B | scratch B | wg | avg grid threads |", "|---|---|---|---|---|---|---|---|"]
    for r in res:
        L.append(f"| `{short(r[0])}` | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]:.0f} |")
    L.append("")
for sub, title in (("pmc_fetch", "FETCH_SIZE (KB; gfx950: x2 for wide coalesced reads, MI355X_MICROARCH.md §HBM)"),
                   ("pmc_write", "WRITE_SIZE (KB)"), ("pmc_valu", "SQ / GRBM counters"), ("pmc_lds", "LDS counters")):
    dbs = glob.glob(os.path.join(out, sub, "*.db"))
    if not dbs:
        continue
    db = sqlite3.connect(dbs[0])
    rows = list(db.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                           "where kernel_name like 'd2::%' or kernel_name like 'void d2::%' group by kernel_name, counter_name"))
    L += [f"## PMC pass: {title}", "", "| kernel | counter | dispatches | sum | avg / dispatch | avg dispatch ns |", "|---|---|---|---|---|---|"]
    for k, c, n, s, a, d in rows:
        L.append(f"| `{short(k)}` | {c} | {n} | {s:.6g} | {a:.6g} | {d:.0f} |")
    L.append("")
# per-launch HBM traffic of the hot kernels from the PMC passes (KB -> bytes).  gfx950: FETCH_SIZE reports half the
# bytes of a wide coalesced read (MI355X_MICROARCH.md §HBM) -> doubled; WRITE_SIZE is taken as reported (uncalibrated).
import json
traffic = {}
for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    dbs = glob.glob(os.path.join(out, sub, "*.db"))
    if not dbs:
        continue
    db = sqlite3.connect(dbs[0])
    # (dispatches shorter than 8 us are the no-op speculative launches of the per-round kernels: not part of the average)
    for k, n, a in db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? and "
                              "(kernel_name like 'd2::%' or kernel_name like 'void d2::%') and "
                              "(duration >= 8000 or (kernel_name not like '%k_nw_ad%' and kernel_name not like '%k_screen%' and kernel_name not like '%k2_screen_multi%' and kernel_name not like '%k3_tail%')) "
                              "group by kernel_name", (cname,)):
        traffic.setdefault(short(k), {})[cname + "_KB_avg"] = a
        traffic[short(k)]["dispatches"] = n
for k, v in traffic.items():
    v["hbm_bytes_per_launch"] = 2 * 1024 * v.get("FETCH_SIZE_KB_avg", 0.0) + 1024 * v.get("WRITE_SIZE_KB_avg", 0.0)
if traffic:
    # the two hot kernels under the names bench.py looks up (profiles/*_traffic_cfgN.json)
    hot = {}
    sc = [v for k, v in traffic.items() if "k2_screen_multi" in k] or [v for k, v in traffic.items() if "k_screen" in k]
    if sc: hot["screen"] = sc[0]
    nw = [v for k, v in traffic.items() if "k_nw_ad" in k or "k_nw_adw" in k]
    if nw: hot["nw"] = max(nw, key=lambda v: v.get("dispatches", 0))
    tl = [v for k, v in traffic.items() if "k3_tail" in k]
    if tl: hot["tail"] = tl[0]
    json.dump({"tag": tag, "command": desc, "screen": hot.get("screen"), "nw": hot.get("nw"), "tail": hot.get("tail"),
               "note": "avg per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 half-count on wide reads)",
               "kernels": traffic}, open(os.path.join(root, f"{tag}_traffic.json"), "w"), indent=1)
open(os.path.join(root, f"{tag}_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
