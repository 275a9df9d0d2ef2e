"""MFMA busy fraction per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (csv).

usage: python scripts/pmc_mfma_util.py counter_collection.csv [out.md]
MfmaUtil (rocprofv3's derived metric, gfx94x formula) = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM).
The csv carries GRBM_GUI_ACTIVE summed over the 8 XCDs, so max is taken as sum / 8; SIMD_NUM = 256 CUs x 4.
The average shader clock during a kernel follows as (GUI_ACTIVE / 8) / duration."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short  # noqa: E402  (the names bench.py's roofline.all_kernels uses)


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    disp = collections.defaultdict(dict)
    for r in rows:
        d = disp[r["Dispatch_Id"]]
        d["name"] = short(r["Kernel_Name"])
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    for d in disp.values():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
            continue
        a = agg[d["name"]]
        a[0] += d["SQ_VALU_MFMA_BUSY_CYCLES"]; a[1] += d["GRBM_GUI_ACTIVE"] / 8.0; a[2] += d["dur"]; a[3] += 1
    lines = ["| kernel | launches | MFMA busy % | avg shader clock GHz | total ms |", "|---|---|---|---|---|"]
    for name, (busy, gui, dur, n) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        if busy <= 0:
            continue
        lines.append(f"| `{name}` | {n} | {100.0 * busy / (gui * 1024.0):.1f} | {gui / dur:.2f} | {dur / 1e6:.2f} |")
    if "--stack" in sys.argv:   # the Generator's forward stack (scripts/bench_gstack.py): one time-weighted figure over its five conv kernels
        stack = [k for k in agg if k.startswith(("conv_up4_kernel", "conv_up3_kernel", "s1t_fwd_rows_kernel"))]
        busy = sum(agg[k][0] for k in stack); gui = sum(agg[k][1] for k in stack); dur = sum(agg[k][2] for k in stack)
        if gui > 0:
            lines.append(f"| **the five forward kernels, time-weighted** | {sum(agg[k][3] for k in stack)} | **{100.0 * busy / (gui * 1024.0):.1f}** | {gui / dur:.2f} | {dur / 1e6:.2f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2 and not sys.argv[2].startswith("--"):
        src = ("python scripts/bench_gstack.py (B = 256, forward launches of the Generator's transposed-conv stack only; apply / pack launches carry no MFMA work and are left out)"
               if "--stack" in sys.argv else None)
        if src:
            open(sys.argv[2], "w").write(out + "\n\nSource: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- " + src + " (counter pass serialises kernels: the clock column is the clock of THIS pass).\n")
            return
        open(sys.argv[2], "w").write(out + "\n\nSource: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python bench.py "
                                     "--steps 2 --warmup 1 --no-cpu-baseline (counter pass serialises kernels; durations are longer than in the timed run).\n")


if __name__ == "__main__":
    main()
