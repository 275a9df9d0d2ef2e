"""HBM-side traffic per launch of the conv kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv).

usage: python scripts/pmc_traffic.py F_counter_collection.csv W_counter_collection.csv out.json
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KB;
on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact
for 16-B streaming stores (the conv epilogue stores of conv_halo.hip are 4..16 B/lane, those of conv_down3 / conv_up3 16 B/lane).
Tags are those of littlegan_amd.ops.Profile (bench.py reports the dominant one)."""
import collections
import csv
import json
import sys


def tag_of(name):
    if "conv_down3_kernel" in name:
        return "conv_igemm_down"
    if "conv_up3_kernel" in name:
        return "conv_igemm_up"
    if "conv_halo_kernel" in name:
        if "conv_halo_kernelIDF16bLi0E" in name or "conv_halo_kernelIfLi0E" in name or "conv_halo_kernel<__bf16, 0" in name or "conv_halo_kernel<float, 0" in name:
            return "conv_igemm_down"
        return "conv_igemm_up"   # MODE 1 (MODE 2, the stride-1 layer, runs in n3_pgemm.hip in the bf16 path)
    if "wgrad_kernel" in name and "n3_" not in name:
        return "wgrad_igemm"
    return None


def collect(path, counter):
    tot, calls = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        t = tag_of(r["Kernel_Name"])
        if t:
            tot[t] += float(r["Counter_Value"]) * 1024.0
            calls[t] += 1
    return tot, calls


def main():
    f, w, out = sys.argv[1:4]
    ft, fc = collect(f, "FETCH_SIZE")
    wt, wc = collect(w, "WRITE_SIZE")
    res = {}
    for t in ft:
        rd = 2.0 * ft[t] / fc[t]
        wr = wt[t] / wc[t] if wc[t] else 0.0
        res[t] = {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "bytes_per_launch": round(rd + wr), "launches_sampled": fc[t]}
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 2 --warmup 11 "
                    "--no-cpu-baseline`; FETCH_SIZE doubled (gfx950 wide-read correction), averages over all launches of the tag")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
