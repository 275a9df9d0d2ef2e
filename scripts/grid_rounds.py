"""Wave quantisation of the step's NON-persistent launches: how many rounds of resident blocks each grid is, and what the last partial
round wastes.  Input: the counter_collection.csv of any `rocprofv3 --pmc <one counter> --output-format csv` pass over bench.py (the csv
carries Grid_Size, Workgroup_Size, LDS_Block_Size, VGPR_Count, Accum_VGPR_Count per dispatch).

usage: python scripts/grid_rounds.py X_counter_collection.csv [CUS=256]
Residency estimate per CU: min(wave slots 32 / waves per block, floor(512 / allocated VGPRs) * 4 / waves per block, 160 KiB / LDS per block).
waste = (ceil(rounds) - rounds) / ceil(rounds) for grids beyond one round: the share of the launch's block-slots that idle in the last round.
A screening aid: dynamic LDS is not in the csv (kernels with `extern __shared__` show 0), persistent kernels size their own grids."""
import collections
import csv
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import parse  # noqa: E402


def main():
    path, cus = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        base, a = parse(r["Kernel_Name"])
        name = base + ("<" + ", ".join(a) + ">" if a else "")
        wg = int(r["Workgroup_Size"]); grid = int(r["Grid_Size"]) // max(wg, 1)
        # rocprofv3 on this image reports HALF the code object's .vgpr_count (apply_kernel 44 vs 86, bwd_apply_kernel<false> 36 vs 72 in the
        # ELF notes): doubled here, i.e. an upper estimate of the allocation
        vg = 2 * (int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0))
        lds = int(r.get("LDS_Block_Size", 0) or 0)
        key = (name, grid, wg, vg, lds)
        rows[key] = rows.get(key, 0) + 1
    print("| kernel | blocks | threads | VGPRs | LDS B | resident / CU | rounds | idle share of the last round | launches |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (name, grid, wg, vg, lds), n in rows.items():
        waves = max(1, wg // 64)
        alloc = max(8, (vg + 7) // 8 * 8)
        per_simd = min(8, 512 // alloc)
        res = min(32 // waves, per_simd * 4 // waves if waves <= 4 else per_simd // ((waves + 3) // 4))
        if lds:
            res = min(res, (160 * 1024) // lds)
        res = max(res, 1)
        rounds = grid / (res * cus)
        waste = 0.0 if rounds <= 1 else (math.ceil(rounds) - rounds) / math.ceil(rounds)
        if rounds > 1 and waste >= 0.05:
            print(f"| `{name[:90]}` | {grid} | {wg} | {vg} | {lds} | {res} | {rounds:.2f} | {100 * waste:.0f} % | {n} |")


if __name__ == "__main__":
    main()
