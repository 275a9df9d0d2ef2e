"""HBM-side traffic per launch of every conv / weight-gradient kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv).

usage: python scripts/pmc_traffic.py F_counter_collection.csv W_counter_collection.csv out.json
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KB;
on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact
for 16-B streaming stores.  Keys are the kernel names of scripts/kernel_names.py = what lg_last_kernel() reports and bench.py
prints in roofline.kernel, so a kernel that is renamed or added shows up under its own key instead of being dropped."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short  # noqa: E402

PREFIXES = ("conv_", "wgrad_", "n3_", "patch_p16", "up_p16", "s1t_fwd")


def collect(path, counter):
    tot, calls = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        t = short(r["Kernel_Name"])
        if t.startswith(PREFIXES):
            tot[t] += float(r["Counter_Value"]) * 1024.0
            calls[t] += 1
    return tot, calls


def main():
    f, w, out = sys.argv[1:4]
    ft, fc = collect(f, "FETCH_SIZE")
    wt, wc = collect(w, "WRITE_SIZE")
    res = {}
    for t in sorted(ft):
        rd = 2.0 * ft[t] / fc[t]
        wr = wt[t] / wc[t] if wc[t] else 0.0
        res[t] = {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "bytes_per_launch": round(rd + wr), "launches_sampled": fc[t]}
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 2 --warmup 11 "
                    "--no-cpu-baseline --no-graph-leg`; FETCH_SIZE doubled (gfx950 wide-read correction), averages over all launches "
                    "of the kernel (all shapes it serves in the step)")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
