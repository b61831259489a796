"""GPU timeline gaps from a rocprofv3 kernel trace (rocpd sqlite): idle time between consecutive kernels, by successor."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
# keep the last full pass only: find last k_final_p
names = [r[0].split('(')[0].replace('d2::','').replace('void ','') for r in rows]
gap_by = collections.defaultdict(float); cnt = collections.Counter(); busy = 0.0
first = None
# restrict to the window between the 2nd and 3rd k_final_p (one steady-state pass)
idx = [i for i,n in enumerate(names) if n.startswith('k_final_p')]
lo, hi = (idx[-2], idx[-1]) if len(idx) >= 2 else (0, len(rows)-1)
for i in range(lo+1, hi+1):
    g = (rows[i][1] - rows[i-1][2]) / 1e3
    key = names[i-1][:18] + ' -> ' + names[i][:18]
    gap_by[key] += max(g, 0); cnt[key] += 1
    busy += (rows[i][2] - rows[i][1]) / 1e3
span = (rows[hi][2] - rows[lo][2]) / 1e3
print(f"pass span {span/1e3:.2f} ms, kernel busy {busy/1e3:.2f} ms, idle {(span-busy)/1e3:.2f} ms")
for k, v in sorted(gap_by.items(), key=lambda kv: -kv[1])[:16]:
    print(f"{k:42s} n={cnt[k]:5d} total {v/1e3:7.3f} ms  avg {v/cnt[k]:7.2f} us")
