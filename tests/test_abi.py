"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads here (no GPU needed) and
exports every symbol include/littlegan_hip.h declares, with the parameter lists the ctypes table uses."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "littlegan_hip.h")


def _protos():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:^|\n)\s*(const char\*|int|size_t)\s+(lg_\w+)\s*\((.*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        plist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        out[name] = (ret, plist)
    return out


def _ctype(decl):
    if "*" in decl:
        return C.c_void_p
    if "long long" in decl:
        return C.c_longlong
    if "size_t" in decl:
        return C.c_size_t
    if "float" in decl:
        return C.c_float
    assert decl.startswith("int "), decl
    return C.c_int


@pytest.fixture(scope="module")
def lib():
    from littlegan_amd.csrc.build import build
    build(verbose=False)
    from littlegan_amd import _lib
    return _lib


def test_header_declares_the_hot_path_entry_points():
    names = set(_protos())
    for n in ("lg_conv2d_s2_fwd", "lg_conv2d_s2_dgrad", "lg_conv2d_s2_wgrad", "lg_convT_s2_fwd", "lg_convT_s2_dgrad",
              "lg_convT_s2_wgrad", "lg_convT_s1_tanh_fwd", "lg_convT_s1_tanh_bwd", "lg_n3_m16_supported", "lg_convT_s1_tanh_fwd_m16", "lg_convT_s1_tanh_bwd_m16", "lg_dense_fwd", "lg_dense_wgrad", "lg_dense_dgrad",
              "lg_heads_fwd_workspace_bytes", "lg_heads_fwd", "lg_heads_dgrad", "lg_heads_wgrad", "lg_instnorm_leaky_stats", "lg_instnorm_leaky_apply",
              "lg_instnorm_leaky_bwd", "lg_instnorm_bwd_db_workspace_bytes", "lg_instnorm_leaky_bwd_db", "lg_bce_heads_loss_fwd_bwd", "lg_l1_tanh_loss_fwd_bwd", "lg_clip_adam_update", "lg_philox4x32", "lg_randn", "lg_augment_workspace_bytes", "lg_augment"):
        assert n in names


def test_library_loads_and_exports_every_declared_symbol(lib):
    handle = lib.load()
    protos = _protos()
    assert set(protos) == set(lib.SIGNATURES), set(protos) ^ set(lib.SIGNATURES)
    for name, (ret, plist) in protos.items():
        assert hasattr(handle, name), f"{name} not exported"
        res, args = lib.SIGNATURES[name]
        assert len(args) == len(plist), f"{name}: ctypes table has {len(args)} params, header {len(plist)}"
        for a, decl in zip(args, plist):
            assert a is _ctype(decl), f"{name}: param '{decl}' bound as {a}"
        exp_ret = {"int": C.c_int, "size_t": C.c_size_t, "const char*": C.c_char_p}[ret]
        assert res is exp_ret
    assert handle.lg_abi_version() == 1


def test_argument_validation_without_gpu(lib):
    """Host-side validation rejects bad arguments before any launch (safe without a GPU)."""
    h = lib.load()
    assert h.lg_conv_pack(None, None, 32, 32, 0, None) == -1
    assert b"null pointer" in h.lg_last_error()
    assert h.lg_dense_fwd(None, None, None, None, 1, 1, 4, None) == -1
    assert h.lg_conv_pack_bytes(64, 128, 0) == 2 * 25 * 64 * 128 * 4
    assert h.lg_conv_pack_bytes(64, 128, 1) == 2 * 25 * 64 * 128 * 2
    # cb == 3: patch down pack [5][npad(cs)][16] + up pack [25][32][cs]
    assert h.lg_conv_pack_bytes(3, 64, 0) == 5 * 64 * 16 * 4 + 25 * 32 * 64 * 4 + 75 * 64 * 4  # + verbatim f32 kernel
    with pytest.raises(lib.LittleGanHipError):
        lib.check(-1, "x")


def test_library_holds_no_packed_fp32_instruction(lib):
    """Round 5: the one build that ever gave launch-to-launch different results lost the low half of a packed fp32
    subtraction (`v_pk_add_f32` on a VGPR pair with a high-register select) when a second wave shared the SIMD.  The library is built
    with -packed-fp32-ops: no `v_pk_{add,mul,fma}_f32` may appear in any gfx950 code object of the product library (disassembled here
    with llvm-objdump, no GPU needed)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import scan_pk_opsel
    if not os.path.exists(scan_pk_opsel.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    tot, per, _ = scan_pk_opsel.scan_library(os.path.join(ROOT, "littlegan_amd", "liblittlegan_hip.so"))
    assert not tot, f"packed fp32 instructions in the product library: {dict(tot)} (kernels with the failing operand form: {len(per)})"
