"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as a rocpd sqlite db.

usage: python scripts/rocpd_stats.py gpurun_out/profNN/x_results.db STEPS [out_prefix]
Writes <out_prefix>.md and <out_prefix>.csv when a prefix is given; always prints the table.
"""
import re
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"| kernel | calls | ms/step | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    csv = ["name,calls,total_ns,avg_ns,min_ns,max_ns,percent"]
    for n, k, t, a, lo, hi in rows:
        s = short(n)
        csv.append(f"\"{s}\",{k},{t},{a:.0f},{lo},{hi},{100 * t / tot:.2f}")
        if 100 * t / tot >= 0.1:
            lines.append(f"| `{s[:110]}` | {k} | {t / 1e6 / steps:.3f} | {a / 1e3:.1f} | {lo / 1e3:.1f} | {hi / 1e3:.1f} | {100 * t / tot:.1f} |")
    lines.append(f"\nkernel time total: {tot / 1e6 / steps:.3f} ms/step over {steps} steps (warmup included)")
    print("\n".join(lines))
    if out:
        open(out + ".md", "w").write("\n".join(lines) + "\n")
        open(out + ".csv", "w").write("\n".join(csv) + "\n")


if __name__ == "__main__":
    main()
