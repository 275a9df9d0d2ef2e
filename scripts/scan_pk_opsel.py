"""Static check of the built library for the packed-fp32 operand form that went wrong on gfx950 in round 5:
a VOP3P fp32 instruction (v_pk_add / mul / fma_f32) whose LOW result selects the HIGH register of a VGPR-pair source (an op_sel
bit set on a source that is v[a:b]).  Disassembles every gfx950 code object of the library (llvm-objdump) and lists the kernels
that contain the form.  `python scripts/scan_pk_opsel.py [library] [--fail]`; exit 1 with --fail if any kernel has it."""
import collections, glob, os, re, shutil, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PAT = re.compile(r"^\s*(v_pk_(?:add|mul|fma)_f32)\s+(.*)$")
OPS = re.compile(r"(v\[\d+:\d+\]|s\[\d+:\d+\]|v\d+|s\d+|vcc|exec|-?[0-9][0-9.xa-fe+-]*)")


def scan_text(lines):
    tot, per, ex, cur = collections.Counter(), collections.Counter(), {}, None
    for line in lines:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            cur = m[1]
            continue
        m = PAT.match(line.split("//")[0])
        if not m:
            continue
        op, rest = m[1], m[2]
        tot[op] += 1
        ms = re.search(r"op_sel:\[([01,]+)\]", rest)
        if not ms:
            continue
        bits = [int(x) for x in ms[1].split(",")]
        srcs = OPS.findall(re.split(r"\s(?:op_sel|neg_lo|neg_hi|clamp)", rest)[0])[1:]
        if any(b and i < len(srcs) and srcs[i].startswith("v[") for i, b in enumerate(bits)):
            per[cur] += 1
            ex.setdefault(cur, " ".join(line.split("//")[0].split()))
    return tot, per, ex


def scan_library(lib):
    tmp = tempfile.mkdtemp(prefix="lgscan")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        tot, per, ex = collections.Counter(), collections.Counter(), {}
        for f in sorted(glob.glob(os.path.join(tmp, "*gfx950"))):
            out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
            t, p, e = scan_text(out.splitlines())
            tot.update(t); per.update(p)
            for k, v in e.items():
                ex.setdefault(k, v)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return tot, per, ex


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "littlegan_amd", "liblittlegan_hip.so")
    tot, per, ex = scan_library(lib)
    print("packed fp32 instructions in the library:", dict(tot))
    print("kernels in which a LOW result reads the HIGH register of a VGPR pair:", len(per))
    for k, v in per.most_common():
        print(f"  {v:5d}  {k[:120]}\n         e.g. {ex[k]}")
    if "--fail" in sys.argv and (per or tot):
        sys.exit(1)


if __name__ == "__main__":
    main()
