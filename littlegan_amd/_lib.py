"""ctypes binding of liblittlegan_hip.so (C ABI: include/littlegan_hip.h).

The product path has NO fallback: if the shared library is missing this module raises at
import of the symbol table, and every op raises `LittleGanHipError` on a non-zero return.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblittlegan_hip.so")

DT_F32, DT_BF16 = 0, 1

P, I, L, F, Z = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t

# name -> (restype, [argtypes])
SIGNATURES = {
    "lg_abi_version": (I, []),
    "lg_last_error": (C.c_char_p, []),
    "lg_conv_pack_bytes": (Z, [I, I, I]),
    "lg_conv_pack": (I, [P, P, I, I, I, P]),
    "lg_conv2d_s2_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "lg_conv_stats_workspace_bytes": (Z, [I, I, I, I, I]),
    "lg_conv_fwd_stats_fused": (I, [I, I, I, I, I, I, I]),
    "lg_conv2d_s2_fwd_stats": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P, Z, P, P]),
    "lg_conv2d_s2_dgrad_m16": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "lg_conv2d_s2_wgrad_m16": (I, [P, P, P, P, P, P, Z, I, I, I, I, I, I, I, P]),
    "lg_conv2d_s2_dgrad": (I, [P, P, P, I, I, I, I, I, I, P]),
    "lg_conv2d_s2_wgrad": (I, [P, P, P, P, Z, I, I, I, I, I, I, I, P]),
    "lg_convT_s2_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "lg_convT_s2_fwd_stats": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P, Z, P, P]),
    "lg_convT_s2_dgrad_m16": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "lg_convT_s2_wgrad_m16": (I, [P, P, P, P, P, P, Z, I, I, I, I, I, I, I, P]),
    "lg_convT_s2_dgrad": (I, [P, P, P, I, I, I, I, I, I, P]),
    "lg_convT_s2_wgrad": (I, [P, P, P, P, Z, I, I, I, I, I, I, I, P]),
    "lg_wgrad_workspace_bytes": (Z, [I, I, I, I, I, I]),
    "lg_convT_s1_tanh_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "lg_convT_s1_tanh_bwd": (I, [P, P, P, P, P, P, P, Z, I, I, I, I, I, I, I, P]),
    "lg_convT_s1_bwd_workspace_bytes": (Z, [I, I, I, I, I, I]),
    "lg_n3_m16_supported": (I, [I, I, I, I, I]),
    "lg_convT_s1_tanh_fwd_m16": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "lg_conv2d_s2_fwd_stats_zn_supported": (I, [I, I, I, I, I, I]),
    "lg_conv2d_s2_fwd_stats_zn": (I, [P, P, F, P, P, P, I, I, I, I, I, I, P, Z, P, P]),
    "lg_convT_s1_tanh_fwd_z16_supported": (I, [I, I, I, I, I]),
    "lg_convT_s1_tanh_fwd_z16": (I, [P, P, F, P, P, P, I, I, I, I, I, I, P]),
    "lg_convT_s1_tanh_bwd_m16": (I, [P, P, P, P, P, P, P, P, P, Z, I, I, I, I, I, I, I, P]),
    "lg_bias_grad_workspace_bytes": (Z, [L, I]),
    "lg_bias_grad": (I, [P, P, P, Z, L, I, I, P]),
    "lg_bias_grad_m16": (I, [P, P, P, P, Z, L, I, I, P]),
    "lg_conv_halo_supported": (I, [I, I, I, I, I, I, I]),
    "lg_instnorm_workspace_bytes": (Z, [I, L]),
    "lg_instnorm_stats_stride": (I, []),
    "lg_instnorm_leaky_stats": (I, [P, P, P, P, P, Z, I, L, I, F, P]),
    "lg_instnorm_leaky_stats_z16": (I, [P, P, P, P, P, Z, I, L, I, F, P, P]),
    "lg_instnorm_stats_finalize": (I, [P, I, P, P, P, I, P]),
    "lg_instnorm_leaky_apply": (I, [P, P, P, P, P, I, L, I, I, F, P]),
    "lg_instnorm_leaky_bwd": (I, [P, P, P, I, P, P, P, P, P, Z, I, L, I, I, F, I, P]),
    "lg_instnorm_bwd_db_workspace_bytes": (Z, [I, L, I]),
    "lg_instnorm_leaky_bwd_db": (I, [P, P, P, I, P, P, P, P, P, I, P, Z, I, L, I, I, F, I, P]),
    "lg_last_kernel": (C.c_char_p, []),
    "lg_clear_kernel": (I, []),
    "lg_instnorm_leaky_apply_z16": (I, [P, P, P, I, P, P, I, L, I, I, F, P]),
    "lg_instnorm_leaky_apply_z16_p": (I, [P, P, I, P, P, P, P, I, P, P, I, L, I, F, P]),
    "lg_instnorm_leaky_bwd_z16": (I, [P, P, P, I, P, P, P, P, P, I, P, Z, I, L, I, I, F, I, P]),
    "lg_instnorm_leaky_bwd_z16_p": (I, [P, P, P, I, P, P, P, P, P, I, P, I, P, Z, I, L, I, I, F, I, P]),
    "lg_conv2d_s2_dgrad_nf": (I, [P, P, P, I, I, I, I, I, P, P, F, P, Z, P, P]),
    "lg_convT_s2_dgrad_nf": (I, [P, P, P, I, I, I, I, I, P, P, F, P, Z, P, P]),
    "lg_convT_s2_dgrad_bn_supported": (I, [I, I, I, I, I]),
    "lg_convT_s2_dgrad_bn": (I, [P, P, P, F, P, P, I, I, I, I, I, P, P, F, P, Z, P, P]),
    "lg_instnorm_bwd_coef": (I, [P, P, I, P, I, L, P]),
    "lg_convT_s1_tanh_bwd_nf": (I, [P, P, P, P, P, P, P, P, Z, I, I, I, I, I, I, I, P, P, F, P, Z, P, P]),
    "lg_dense_fwd": (I, [P, P, P, P, I, I, I, P]),
    "lg_dense_wgrad": (I, [P, P, P, P, I, I, I, I, P]),
    "lg_dense_dgrad": (I, [P, P, P, I, I, I, P]),
    "lg_concat_cols": (I, [P, I, P, I, P, I, P]),
    "lg_adj_conditions": (I, [P, P, P, P, I, I, P]),
    "lg_heads_fwd_workspace_bytes": (Z, [I, I, I]),
    "lg_heads_fwd": (I, [P, P, P, P, P, P, P, Z, I, I, I, P]),
    "lg_heads_dgrad": (I, [P, P, P, P, I, I, I, P]),
    "lg_heads_wgrad": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "lg_bce_heads_loss_fwd_bwd": (I, [P, P, F, F, F, P, P, I, I, I, P]),
    "lg_l1_workspace_bytes": (Z, []),
    "lg_l1_tanh_loss_fwd_bwd": (I, [P, P, P, P, P, P, Z, L, F, I, P]),
    "lg_clip_adam_update": (I, [P, P, P, P, L, P, F, F, F, F, F, F, P]),
    "lg_adam_advance": (I, [P, F, F, P]),
    "lg_axpby": (I, [P, P, F, F, L, P]),
    "lg_philox4x32": (I, [P, I, L, L, P]),   # seed / offset: unsigned long long in C, passed as their 64-bit pattern
    "lg_randn": (I, [P, L, F, F, L, L, P]),
    "lg_augment_workspace_bytes": (Z, [I]),
    "lg_augment": (I, [P, P, I, I, I, P, F, F, F, F, L, L, P, Z, P]),
    "lg_fid_stats_workspace_bytes": (Z, [L, I]),
    "lg_fid_stats": (I, [P, L, I, P, P, P, Z, P]),
    "lg_augment_drawn_workspace_bytes": (Z, [I]),
    "lg_device_cus": (I, []),
    "lg_set_reserved_cus": (I, [I]),
    "lg_grid_cus": (I, []),
    "lg_contention_probe": (I, [P, P, L, I, I, I, P]),
    "lg_set_clock_census": (I, [P]),
    "lg_clock_sample": (I, [P, I, P]),
    "lg_augment_drawn": (I, [P, P, I, I, I, F, F, F, F, F, L, L, L, P, Z, P]),
}


class LittleGanHipError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the HIP library (once).  Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LittleGanHipError(
            f"{LIB_PATH} not found: build it with `python -m littlegan_amd.csrc.build` "
            "(the LittleGAN hot path has no CPU fallback)")
    # Probe / ablation variants (scripts/probe/*: LG_EXTRA_FLAGS macros, "results wrong, timing only") live in their own
    # liblittlegan_hip_<variant>.so and are loaded ONLY when LG_LIB_VARIANT names one; the product library must have been
    # linked with the default flags.
    from .csrc import build as _build
    variant = os.environ.get("LG_LIB_VARIANT") or None
    path = _build.variant_paths(variant)[1]
    if variant is None:
        fl = _build.built_flags()
        if fl is not None and fl != " ".join(_build.FLAGS):
            raise LittleGanHipError(f"{LIB_PATH} was built with non-default flags ({fl!r}); rebuild with `python -m littlegan_amd.csrc.build`")
    elif not os.path.exists(path):
        raise LittleGanHipError(f"{path} not found: build it with LG_EXTRA_FLAGS=... python -m littlegan_amd.csrc.build --variant {variant}")
    # torch first: it ships its own libamdhip64 and must be the HIP runtime of the process.  If this library were loaded
    # before torch, the system runtime it links against would come in as a SECOND runtime and its kernels would see no device.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.lg_abi_version() != 1:
        raise LittleGanHipError("liblittlegan_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().lg_last_error().decode("utf-8", "replace")
        raise LittleGanHipError(f"{what} failed (rc={rc}): {msg}")
