"""tools/pmc_busy.py <summary.md> - derived issue figures from the SQ_* rows of a profiles/summarize.py summary (the PMC pass with
SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES, and the LDS
pass): per kernel, summed over its dispatches,
  valu_of_wave_cycles = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (share of a resident wave's cycles in which it issues a VALU instruction;
                        x resident waves per SIMD = the SIMD's VALU-busy share, 1.0 = the 78.6 T int-ops/s the rooflines are quoted against)
  waiting             = SQ_WAIT_ANY / SQ_WAVE_CYCLES           (share of wave cycles spent waiting on memory / LDS / barriers)
  lds_conflict        = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
Prints a markdown table."""
import re
import sys
from collections import defaultdict

rows = defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"\| `(.+?)` \| (SQ_[A-Z_]+) \| (\d+) \| ([0-9.e+]+) \| ([0-9.e+]+) \| (\d+) \|", line)
    if m:
        rows[m.group(1)][m.group(2)] = (float(m.group(4)), int(m.group(3)), int(m.group(6)))
print("| kernel | dispatches | avg ns | VALU instr / wave-cycle | waiting / wave-cycle | waves per dispatch | LDS bank-conflict cycles / LDS active cycles |")
print("|---|---|---|---|---|---|---|")
for k, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0,))[0]):
    if "SQ_WAVE_CYCLES" not in c or c["SQ_WAVE_CYCLES"][0] < 1e8:
        continue
    wc = c["SQ_WAVE_CYCLES"][0]
    lds = (c["SQ_LDS_BANK_CONFLICT"][0] / c["SQ_LDS_IDX_ACTIVE"][0]) if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", (0,))[0] > 0 else float("nan")
    print(f"| `{k}` | {c['SQ_WAVE_CYCLES'][1]} | {c['SQ_WAVE_CYCLES'][2]} | {c['SQ_ACTIVE_INST_VALU'][0] / wc:.3f} | {c['SQ_WAIT_ANY'][0] / wc:.3f} | "
          f"{c['SQ_WAVES'][0] / c['SQ_WAVES'][1]:.0f} | {lds:.3f} |")
