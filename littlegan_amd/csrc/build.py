"""Builds liblittlegan_hip.so (gfx950) in-tree with hipcc.  `python -m littlegan_amd.csrc.build`"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["capi.hip", "conv_igemm.hip", "conv_halo.hip", "conv_down3.hip", "conv_up3.hip", "n3_kernels.hip", "n3_pgemm.hip", "wgrad_igemm.hip", "pack.hip", "norm.hip", "dense.hip", "heads.hip", "loss_optim.hip", "augment.hip", "fid.hip", "wgrad_at.hip", "n3_rows.hip", "skinny_mfma.hip", "conv_up4.hip"]
LIB = os.path.join(PKG, "liblittlegan_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(HERE, "lg_common.h"), os.path.join(os.path.dirname(PKG), "include", "littlegan_hip.h")]

    def cc(src):
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + os.environ.get("LG_EXTRA_FLAGS", "").split() + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
