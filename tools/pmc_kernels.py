"""Per-kernel averages of the counters in a rocprofv3 --pmc run (rocpd sqlite).  usage: pmc_kernels.py <db> [min_duration_ns]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
mind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                  "where duration >= ? group by kernel_name, counter_name", (mind,))
tab = collections.defaultdict(dict)
for k, c, n, a, d in rows:
    k = k.replace("d2::", "").split("(")[0][:40]
    tab[k][c] = a; tab[k]["_n"] = n; tab[k]["_us"] = d / 1e3
cols = sorted({c for v in tab.values() for c in v if not c.startswith("_")})
print("| kernel | n | avg us | " + " | ".join(cols) + " |")
print("|---|---|---|" + "---|" * len(cols))
for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["_us"] * kv[1]["_n"]):
    print(f"| `{k}` | {v['_n']} | {v['_us']:.1f} | " + " | ".join(f"{v.get(c, 0):.4g}" for c in cols) + " |")
