"""Builds liblittlegan_hip.so (gfx950) in-tree with hipcc.  `python -m littlegan_amd.csrc.build`"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["capi.hip", "conv_igemm.hip", "conv_halo.hip", "conv_down3.hip", "conv_up3.hip", "n3_kernels.hip", "n3_pgemm.hip", "wgrad_igemm.hip", "pack.hip", "norm.hip", "dense.hip", "heads.hip", "loss_optim.hip", "augment.hip", "fid.hip", "wgrad_at.hip", "wgrad_at32.hip", "n3_rows.hip", "skinny_mfma.hip", "conv_up4.hip", "runtime.hip"]
LIB = os.path.join(PKG, "liblittlegan_hip.so")
# -packed-fp32-ops (round 5): no v_pk_{add,mul,fma}_f32 anywhere in the library.  The one kernel build that ever gave launch-to-launch
# different results lost the low half of a packed fp32 subtraction (VGPR pair, high-register select) with a second wave on the SIMD; without
# packed fp32 instructions the same source is deterministic, and the whole library is as fast (C3 10.944 / 10.948 against 10.955 / 10.958 ms,
# C2 8.912 / 8.906, C5 39.89 / 39.96: scripts/probe/nopk_ab.sh) — the instruction class is simply not generated.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def variant_paths(variant):
    """A probe / ablation VARIANT (LG_EXTRA_FLAGS macros) is built beside the product library, never over it:
    objects in csrc/build_<variant>/, library liblittlegan_hip_<variant>.so; loaded only when LG_LIB_VARIANT names it."""
    if not variant:
        return os.path.join(HERE, "build"), LIB
    return os.path.join(HERE, "build_" + variant), os.path.join(PKG, f"liblittlegan_hip_{variant}.so")


def build(force=False, verbose=True, variant=None):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir, LIB = variant_paths(variant)
    if not variant and os.environ.get("LG_EXTRA_FLAGS"):
        raise SystemExit("LG_EXTRA_FLAGS needs --variant NAME: ablation builds never replace the product library")
    os.makedirs(objdir, exist_ok=True)
    common_h, public_h = os.path.join(HERE, "lg_common.h"), os.path.join(os.path.dirname(PKG), "include", "littlegan_hip.h")

    def hdrs_of(path):   # the public header is a dependency of the sources that include it (capi.hip, runtime.hip), not of every kernel file
        return [common_h] + ([public_h] if "littlegan_hip.h" in open(path).read() else [])

    # LG_EXTRA_FLAGS carries the ablation macros of scripts/probe/*.sh ("results wrong, timing only"): the flag set an object
    # was built with is recorded beside it, and an object (hence the library) built with OTHER flags is stale — a probe build
    # can never be picked up silently by the next test / bench / training run.
    flags = FLAGS + os.environ.get("LG_EXTRA_FLAGS", "").split()
    flag_line = " ".join(flags)
    # a variant may name the sources its macros touch (LG_VARIANT_SOURCES=conv_down3.hip,...): only those are compiled with the
    # extra flags, every other object is the product build's (seconds instead of minutes per variant)
    only = [s_ for s_ in os.environ.get("LG_VARIANT_SOURCES", "").split(",") if s_] if variant else []
    base_objdir = os.path.join(HERE, "build")

    def cc(src):
        s = os.path.join(HERE, src)
        if only and src not in only:
            o = os.path.join(base_objdir, src.replace(".hip", ".o"))
            if not os.path.exists(o):
                raise SystemExit(f"{o} missing: build the product library first")
            return o
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        fl = o + ".flags"
        same_flags = os.path.exists(fl) and open(fl).read() == flag_line
        if force or not same_flags or _stale(o, [s] + hdrs_of(s)):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            if os.path.exists(fl):
                os.remove(fl)
            subprocess.run(cmd, check=True)
            with open(fl, "w") as f:
                f.write(flag_line)
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, SOURCES))
    lib_fl = os.path.join(objdir, "lib.flags")
    if force or _stale(LIB, objs) or not (os.path.exists(lib_fl) and open(lib_fl).read() == flag_line):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(lib_fl, "w") as f:
            f.write(flag_line)
    return LIB


def built_flags(variant=None):
    """The flag line the in-tree library was linked from (None if unknown) — _lib.load() refuses a probe build."""
    p = os.path.join(variant_paths(variant)[0], "lib.flags")
    return open(p).read() if os.path.exists(p) else None


if __name__ == "__main__":
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print("built", build(force="--force" in sys.argv, variant=var))
