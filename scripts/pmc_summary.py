"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (short name) sum of each counter and calls."""
import csv, sys, re, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r.get("Kernel_Name") or r.get("Kernel Name")
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)[:70]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(name, r["Counter_Name"])] += 1
ctrs = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(72), " ".join(c[:22].rjust(22) for c in ctrs))
for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1].values()))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(k.ljust(72), " ".join(("%.4g" % v.get(c, 0)).rjust(22) for c in ctrs), " calls", max(calls[(k, c)] for c in ctrs))
