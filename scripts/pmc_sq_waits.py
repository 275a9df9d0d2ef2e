"""Wave-state and LDS-bank summary per kernel from a rocprofv3 --pmc pass (csv) collecting
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE.
usage: python scripts/pmc_sq_waits.py counter_collection.csv [out.md]"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
seen = set()
for r in rows:
    n = short(r["Kernel_Name"])
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        dur[n] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
lines = ["| kernel | parked (s_waitcnt / barrier) % | issue stall % | issuing % | LDS active % of wave cycles | bank-conflict % of LDS cycles | ms |",
         "|---|---|---|---|---|---|---|"]
for n, d in sorted(dur.items(), key=lambda kv: -kv[1])[:16]:
    a = agg[n]
    wc = a["SQ_WAVE_CYCLES"] or 1
    lines.append(f"| `{n}` | {100 * a['SQ_WAIT_ANY'] / wc:.0f} | {100 * a['SQ_WAIT_INST_ANY'] / wc:.0f} | {100 * a['SQ_ACTIVE_INST_ANY'] / wc:.0f} | "
                 f"{100 * a['SQ_LDS_IDX_ACTIVE'] / wc:.1f} | {100 * a['SQ_LDS_BANK_CONFLICT'] / (a['SQ_LDS_IDX_ACTIVE'] or 1):.0f} | {d / 1e6:.2f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n\nFractions of SQ_WAVE_CYCLES (wave-resident cycles).  Source: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY "
                                 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline\n")
