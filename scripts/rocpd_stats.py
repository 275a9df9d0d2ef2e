"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as a rocpd sqlite db.

usage: python scripts/rocpd_stats.py gpurun_out/profNN/x_results.db STEPS [out_prefix]
Writes <out_prefix>.md and <out_prefix>.csv when a prefix is given; always prints the table.
"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import parse  # noqa: E402


def short(n):
    """full template instance, spelled uniformly (rocprofv3 mixes demangled / mangled / botched-__bf16 names)"""
    base, a = parse(n)
    return base + ("<" + ", ".join(a) + ">" if a else "")


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else None
    c = sqlite3.connect(db)
    raw = c.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()
    merged = {}
    for n, k, t, lo, hi in raw:   # two spellings of one instance collapse into one row
        m = merged.setdefault(short(n), [0, 0, 1 << 62, 0])
        m[0] += k; m[1] += t; m[2] = min(m[2], lo); m[3] = max(m[3], hi)
    rows = sorted(((n, k, t, t / k, lo, hi) for n, (k, t, lo, hi) in merged.items()), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    lines = [f"| kernel | calls | ms/step | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    csv = ["name,calls,total_ns,avg_ns,min_ns,max_ns,percent"]
    for n, k, t, a, lo, hi in rows:
        s = n
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
