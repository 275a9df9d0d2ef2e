"""rocprofv3 kernel names -> the short kernel-template names the C side reports through lg_last_kernel() (and bench.py prints in
`roofline.kernel` / `roofline.all_kernels`), so that the HIP-event view (bench.py), the kernel trace (rocpd_stats.py) and the
counter passes (pmc_*.py) speak about the same kernels.

rocprofv3 hands out three spellings of one template instance: demangled ("conv_up3_kernel<64, 32, true, false>"), still mangled
("_ZN12_GLOBAL__N_116conv_halo_kernelIDF16bLi0ELi2ELb0E...") and — for __bf16 template arguments — a botched demangling
("conv_halo_kernel<bool _Accum, int, E, 4, false, ...>": `DF16b Li1E` read as `bool _Accum, int, E`)."""
import re


def _mangled_args(s):
    """template arguments of an Itanium-mangled instance made of types (f, DF16b) and literals (LiNE, LbNE) only"""
    out, i = [], 0
    while i < len(s) and s[i] != "E":
        if s.startswith("DF16b", i):
            out.append("__bf16"); i += 5
        elif s[i] == "f":
            out.append("float"); i += 1
        elif s[i] == "L":
            m = re.match(r"L([ib])(n?\d+)E", s[i:])
            if not m:
                break
            out.append(("true" if m.group(2) != "0" else "false") if m.group(1) == "b" else m.group(2).replace("n", "-"))
            i += m.end()
        else:
            break
    return out


def parse(name):
    """-> (base kernel name, [template arguments as strings]); arguments [] when there are none / unknown"""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").strip()
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:
        ln = int(m.group(1))
        base = n[m.end():m.end() + ln]
        rest = n[m.end() + ln:]
        return base, (_mangled_args(rest[1:]) if rest.startswith("I") else [])
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"([A-Za-z_][\w:]*)<(.*)>$", n)
    if not m:
        return n, []
    args = [a.strip().replace("(bool)", "").replace("(int)", "") for a in m.group(2).split(",")]
    if args[:3] == ["bool _Accum", "int", "E"]:   # botched `DF16b, Li<MODE>E`: the type is __bf16, the MODE literal is lost
        args = ["__bf16", "?"] + args[3:]
    return m.group(1), args


def short(name):
    base, a = parse(name)
    t = lambda i, d=None: a[i] if i < len(a) else d
    if base == "conv_down3_kernel":      # <STATS, FUSE, PAIR = false, NW = 128, NORM = 0>; NORM: 0 none, 1 forward norm, 2 BWDNORM (round 3: a bool)
        if t(2) == "true":
            return "conv_down3_kernel<PAIR>"
        norm = {"true": ",NORM", "1": ",NORM", "2": ",BWDNORM"}.get(t(4, "0"), "")
        return f"conv_down3_kernel<NW={t(3, '128')}{norm}>"
    if base == "conv_up3_kernel":        # <CS, N, STATS, FUSE, NTT = tiles per step>; (64, 32) with one tile per step = the 4-wave form
        return f"conv_up3_kernel<{t(0)},{t(1)}" + (",4w>" if (t(1) == "32" and t(4, "2") == "1") else ">")
    if base == "conv_halo_kernel":       # <T, MODE, KCH, DBUF, SRC16, RES, ...>
        ty = "bf16" if t(0) == "__bf16" else "f32"
        mode = t(1)
        if mode == "?":                  # K-sliced UP tiles are the KCH = 4 builds, DOWN tiles KCH = 2 (conv_halo.hip: halo_w3_ok)
            mode = "1" if t(2) == "4" else "0"
        if mode == "0":
            return f"conv_halo_kernel<{ty},DOWN>"
        if mode == "1":
            if t(5) == "true":
                return "conv_halo_kernel<UP,resident>" if ty == "bf16" else "conv_halo_kernel<f32,UP,resident>"
            return f"conv_halo_kernel<{ty},UP,K-sliced>"
        return "conv_halo_kernel<S1T>"
    if base == "conv_igemm_kernel":      # <T, MODE, ...>
        return "conv_igemm_kernel<%s>" % {"0": "DOWN", "1": "UP", "2": "S1T", "3": "PATCH"}.get(t(1), "?")
    if base == "wgrad_at_kernel":
        return f"wgrad_at_kernel<{t(0)},{t(1)}>"
    if base == "wgrad_at32_kernel":
        return f"wgrad_at32_kernel<{t(0)},{t(1)}>"
    if base == "wgrad_kernel":           # <BF16, PATCH, SRC16, ...>
        return "wgrad_kernel<PATCH>" if t(1) == "true" else ("wgrad_kernel<bf16,per-tap>" if t(0) == "true" else "wgrad_kernel<f32,per-tap>")
    if base == "n3_wgrad16_kernel":      # <NT, TH = 8>
        return f"n3_wgrad16_kernel<{t(0)}" + (",16>" if t(1) == "16" else ">")
    if base == "n3_wgrad_kernel":
        return "n3_wgrad_kernel<f32>"
    if base == "patch_p16_kernel":       # <S, N, OUT16, STATS, NF = false>
        return f"patch_p16_kernel<{t(0)},{t(1)}" + (",nf>" if t(4) == "true" else ">")
    if base == "s1t_fwd_rows_kernel":
        return "s1t_fwd_rows_kernel<32,NORM>" if t(1) == "true" else "s1t_fwd_rows_kernel<32>"
    if base == "conv_up4_kernel":        # <STATS, FUSE, PAIR = false>
        return "conv_up4_kernel<PAIR>" if t(2) == "true" else "conv_up4_kernel"
    if base in ("up_p16_kernel", "s1t_fwd_p16_kernel"):
        return base
    return base + ("<" + ",".join(a) + ">" if a else "")
