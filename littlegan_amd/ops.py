"""torch-tensor front end of the C ABI (include/littlegan_hip.h).

PyTorch here is plumbing only: it owns device memory and the HIP stream.  Every function
validates shapes on the host (a wrong shape must never reach a hand-written kernel), then
enqueues the HIP kernels on torch's current stream.  No fallback paths.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import DT_BF16, DT_F32, check

_WS = {}
_WS_RETIRED = []     # buffers replaced by a bigger one after a HIP graph captured their address: kept alive for the replays
_WS_PINNED = False   # set by pin_workspaces() (EagerTrainer.graph_step) as soon as any graph holds workspace pointers
NSTAT = 8  # floats per sample in the instance-norm statistics record (lg_instnorm_stats_stride)


class Profile:
    """Optional per-launch HIP-event timing of the conv contractions (bench.py's live roofline leg).
    Events are recorded on torch's current stream = the stream the kernels are launched on."""
    enabled = False
    records = []  # (tag, algorithmic flops, start event, end event)
    after_conv = None  # bench.py's clock leg: called (no arguments) right behind every timed conv-class launch, on the launch stream

    @classmethod
    def start(cls):
        cls.records = []
        cls.enabled = True

    @classmethod
    def stop(cls):
        """-> {tag: (launches, total flops, total seconds)} ; call after a device synchronize"""
        cls.enabled = False
        out = {}
        for tag, fl, e0, e1 in cls.records:
            n, f, t = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, f + fl, t + e0.elapsed_time(e1) * 1e-3)
        cls.records = []
        return out


def _pb():
    if not Profile.enabled:
        return None
    _lib.load().lg_clear_kernel()   # lg_last_kernel is sticky: a launch path that names no kernel must not inherit the previous call's
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _pe(e0, tag, flops):
    """tag = class of the contraction (what the FLOP formula depends on) + the kernel template the C side really launched"""
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        kern = _lib.load().lg_last_kernel().decode() or tag   # unnamed launch path: filed under its class tag
        Profile.records.append((f"{tag}:{kern}", float(flops), e0, e1))
        if Profile.after_conv is not None:
            Profile.after_conv()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, shape=None, name="tensor"):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: need a contiguous fp32 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
    return t


def pin_workspaces():
    """Call before a HIP-graph capture: a captured graph holds the RAW addresses of the scratch buffers below, so from now on
    a buffer that is outgrown is retired (kept allocated) instead of being handed back to the caching allocator, where a
    later tensor could land under a replaying graph's writes."""
    global _WS_PINNED
    _WS_PINNED = True


def last_kernel() -> str:
    """Name of the kernel the most recent conv-class C-ABI call launched on this thread (lg_last_kernel)."""
    return _lib.load().lg_last_kernel().decode()


def workspace(nbytes: int, device, tag="default") -> torch.Tensor:
    """Grow-only scratch buffer per (device, tag); kernels on one stream run in order so sharing is safe."""
    key = (str(device), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and _WS_PINNED:
            _WS_RETIRED.append(buf)
        if torch.cuda.is_current_stream_capturing():
            raise _lib.LittleGanHipError(f"workspace '{tag}' must grow to {nbytes} bytes during a HIP-graph capture: run the "
                                         "step once eagerly at this shape first")
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ------------------------------------------------------------------ conv layers
def conv_pack_bytes(cb, cs, dtype):
    return int(_lib.load().lg_conv_pack_bytes(cb, cs, dtype))


def conv_pack(w, cb, cs, dtype, out=None):
    """w: master kernel [5,5,cb,cs] -> packed MFMA operand images (uint8 buffer)."""
    _chk(w, (5, 5, cb, cs), "w")
    if out is None:
        out = torch.empty(conv_pack_bytes(cb, cs, dtype), dtype=torch.uint8, device=w.device)
    check(_lib.load().lg_conv_pack(_p(w), _p(out), cb, cs, dtype, _stream()), "lg_conv_pack")
    return out


def conv2d_s2_fwd(x, pack, bias, cs, dtype, out=None):
    B, H, W, cb = x.shape
    _chk(x, name="x")
    if H % 2 or W % 2:
        raise ValueError("conv2d_s2_fwd: H and W must be even")
    if out is None:
        out = torch.empty(B, H // 2, W // 2, cs, dtype=torch.float32, device=x.device)
    _chk(out, (B, H // 2, W // 2, cs), "out")
    if bias is not None:
        _chk(bias, (cs,), "bias")
    e0 = _pb()
    check(_lib.load().lg_conv2d_s2_fwd(_p(x), _p(pack), _p(bias), _p(out), B, H // 2, W // 2, cb, cs, dtype, _stream()),
          "lg_conv2d_s2_fwd")
    _pe(e0, "conv_igemm_patch" if cb == 3 else "conv_igemm_down", 50.0 * B * (H // 2) * (W // 2) * cb * cs)
    return out


class NormPartials:
    """First-pass sums of an InstanceNormalization backward, produced by the epilogue of the conv that wrote the gradient
    (lg_*_dgrad_nf): `buf` holds [B][nparts][2] doubles.  Valid until the next fused data gradient is enqueued (one
    shared workspace; kernels of a stream run in order and the consumer is the very next norm backward)."""

    def __init__(self, buf, nparts, alpha, shape):
        self.buf, self.nparts, self.alpha, self.shape = buf, nparts, float(alpha), tuple(shape)


def _nf_args(fuse, B, up, Hs, Ws, N, device):
    """fuse = (z16, stats, alpha) of the layer the produced gradient belongs to -> ctypes arguments + result holder"""
    import ctypes
    z16, st, alpha = fuse
    if z16.dtype != torch.bfloat16 or not z16.is_contiguous() or z16.shape[0] != B:
        raise ValueError("fuse: z16 must be the contiguous bf16 conv output of the layer below")
    _chk(st, (B, NSTAT), "fuse stats")
    lib = _lib.load()
    ws = workspace(int(lib.lg_conv_stats_workspace_bytes(int(up), B, Hs, Ws, N)), device, "nfpart")
    return z16, st, float(alpha), ws, ctypes.c_int(0)


def conv2d_s2_dgrad(dy, pack, cb, dtype, out=None, dy16=None, out_bf16=False, fuse=None):
    """dx [B,2Hs,2Ws,cb]; out_bf16: return the gradient as a bf16 tensor (no fp32 copy is written).
    dy may be None when its bf16 mirror dy16 is given and the halo kernel covers the shape (conv_halo_supported).
    fuse = (z16, stats, alpha) (bf16 path, out_bf16): the kernel also writes the first-pass sums of the norm backward the
    gradient feeds; returns (dx, NormPartials or None)."""
    B, Hs, Ws, cs = (dy if dy is not None else dy16).shape
    if fuse is not None:
        import ctypes
        if not (out_bf16 and dy16 is not None and dtype == DT_BF16):
            raise ValueError("conv2d_s2_dgrad: fuse needs the bf16 path (dy16, out_bf16)")
        _chk16(dy16, dy16, "dy16")
        out = torch.empty(B, 2 * Hs, 2 * Ws, cb, dtype=torch.bfloat16, device=dy16.device)
        z16, st, alpha, ws, npo = _nf_args(fuse, B, True, Hs, Ws, cb, dy16.device)
        e0 = _pb()
        check(_lib.load().lg_conv2d_s2_dgrad_nf(_p(dy16), _p(pack), _p(out), B, Hs, Ws, cb, cs, _p(z16), _p(st), alpha, _p(ws),
                                               ws.numel(), ctypes.addressof(npo), _stream()), "lg_conv2d_s2_dgrad_nf")
        _pe(e0, "conv_igemm_up", 50.0 * B * Hs * Ws * cb * cs)
        return out, (NormPartials(ws, npo.value, alpha, out.shape) if npo.value > 0 else None)
    if dy is not None:
        _chk(dy, name="dy")
    if dy16 is not None:
        _chk16(dy16, dy if dy is not None else dy16, "dy16")
    if dtype != DT_BF16 and (dy is None or out_bf16):
        raise ValueError("conv2d_s2_dgrad: dtype f32 needs the fp32 gradient and writes fp32 (bf16 mirror / bf16 output: dtype bf16)")
    dev = (dy if dy is not None else dy16).device
    if out_bf16:
        out = torch.empty(B, 2 * Hs, 2 * Ws, cb, dtype=torch.bfloat16, device=dev)
        o32, o16 = None, out
    else:
        if out is None:
            out = torch.empty(B, 2 * Hs, 2 * Ws, cb, dtype=torch.float32, device=dev)
        _chk(out, (B, 2 * Hs, 2 * Ws, cb), "out")
        o32, o16 = out, None
    e0 = _pb()
    check(_lib.load().lg_conv2d_s2_dgrad_m16(_p(dy), _p(dy16), _p(pack), _p(o32), _p(o16), B, Hs, Ws, cb, cs, dtype, _stream()),
          "lg_conv2d_s2_dgrad_m16")
    _pe(e0, "conv_igemm_up_n3" if cb == 3 else "conv_igemm_up", 50.0 * B * Hs * Ws * cb * cs)
    return out


def _wgrad(fn_name, big, small, dw, accumulate, dtype, swap, big16=None, small16=None):
    lib = _lib.load()
    B, Hs, Ws, cs = (small if small is not None else small16).shape
    cb = (big if big is not None else big16).shape[3]
    if big is not None:
        _chk(big, (B, 2 * Hs, 2 * Ws, cb), "big")
    if small is not None:
        _chk(small, name="small")
    n3 = cb == 3 and big is not None and small16 is not None  # 3-channel layer: fp32 image + mirror of the wide operand
    if (big is None or small is None) and (big16 is None or small16 is None) and not n3:
        raise ValueError("wgrad: an fp32 operand may be omitted only when BOTH bf16 mirrors are given")
    _chk(dw, (5, 5, cb, cs), "dw")
    nbytes = int(lib.lg_wgrad_workspace_bytes(B, Hs, Ws, cb, cs, dtype))
    ws = workspace(nbytes, dw.device, "wgrad")
    if n3:
        big16 = None
        _chk16(small16, small if small is not None else small16, "small16")
    elif big16 is None or small16 is None:
        big16 = small16 = None  # the bf16-source kernel needs both mirrors
    else:
        _chk16(big16, big if big is not None else big16, "big16")
        _chk16(small16, small if small is not None else small16, "small16")
    (a, a16), (b, b16) = ((small, small16), (big, big16)) if swap else ((big, big16), (small, small16))
    e0 = _pb()
    check(getattr(lib, fn_name)(_p(a), _p(a16), _p(b), _p(b16), _p(dw), _p(ws), ws.numel(), B, Hs, Ws, cb, cs,
                                int(accumulate), dtype, _stream()), fn_name)
    _pe(e0, "wgrad_patch" if cb == 3 else "wgrad_igemm", 50.0 * B * Hs * Ws * cb * cs)
    return dw


def conv2d_s2_wgrad(x, dy, dw, accumulate, dtype, x16=None, dy16=None):
    """dw[5,5,cb,cs] (+)= wgrad(x [B,2Hs,2Ws,cb], dy [B,Hs,Ws,cs]); x16/dy16: optional bf16 mirrors"""
    return _wgrad("lg_conv2d_s2_wgrad_m16", x, dy, dw, accumulate, dtype, swap=False, big16=x16, small16=dy16)


def convT_s2_fwd(x, pack, bias, cb, dtype, out=None):
    B, Hs, Ws, cs = x.shape
    _chk(x, name="x")
    if out is None:
        out = torch.empty(B, 2 * Hs, 2 * Ws, cb, dtype=torch.float32, device=x.device)
    _chk(out, (B, 2 * Hs, 2 * Ws, cb), "out")
    if bias is not None:
        _chk(bias, (cb,), "bias")
    e0 = _pb()
    check(_lib.load().lg_convT_s2_fwd(_p(x), _p(pack), _p(bias), _p(out), B, Hs, Ws, cb, cs, dtype, _stream()),
          "lg_convT_s2_fwd")
    _pe(e0, "conv_igemm_up", 50.0 * B * Hs * Ws * cb * cs)
    return out


def convT_s2_dgrad(dy, pack, cs, dtype, out=None, dy16=None, out_bf16=False, fuse=None):
    """fuse: as conv2d_s2_dgrad -> returns (dx, NormPartials or None)"""
    B, H, W, cb = (dy if dy is not None else dy16).shape
    if fuse is not None:
        import ctypes
        if not (out_bf16 and dy16 is not None and dtype == DT_BF16):
            raise ValueError("convT_s2_dgrad: fuse needs the bf16 path (dy16, out_bf16)")
        _chk16(dy16, dy16, "dy16")
        out = torch.empty(B, H // 2, W // 2, cs, dtype=torch.bfloat16, device=dy16.device)
        z16, st, alpha, ws, npo = _nf_args(fuse, B, False, H // 2, W // 2, cs, dy16.device)
        e0 = _pb()
        check(_lib.load().lg_convT_s2_dgrad_nf(_p(dy16), _p(pack), _p(out), B, H // 2, W // 2, cb, cs, _p(z16), _p(st), alpha,
                                              _p(ws), ws.numel(), ctypes.addressof(npo), _stream()), "lg_convT_s2_dgrad_nf")
        _pe(e0, "conv_igemm_down", 50.0 * B * (H // 2) * (W // 2) * cb * cs)
        return out, (NormPartials(ws, npo.value, alpha, out.shape) if npo.value > 0 else None)
    if dy is not None:
        _chk(dy, name="dy")
    if dy16 is not None:
        _chk16(dy16, dy if dy is not None else dy16, "dy16")
    if dtype != DT_BF16 and (dy is None or out_bf16):
        raise ValueError("convT_s2_dgrad: dtype f32 needs the fp32 gradient and writes fp32 (bf16 mirror / bf16 output: dtype bf16)")
    dev = (dy if dy is not None else dy16).device
    if out_bf16:
        out = torch.empty(B, H // 2, W // 2, cs, dtype=torch.bfloat16, device=dev)
        o32, o16 = None, out
    else:
        if out is None:
            out = torch.empty(B, H // 2, W // 2, cs, dtype=torch.float32, device=dev)
        _chk(out, (B, H // 2, W // 2, cs), "out")
        o32, o16 = out, None
    e0 = _pb()
    check(_lib.load().lg_convT_s2_dgrad_m16(_p(dy), _p(dy16), _p(pack), _p(o32), _p(o16), B, H // 2, W // 2, cb, cs, dtype,
                                            _stream()), "lg_convT_s2_dgrad_m16")
    _pe(e0, "conv_igemm_down", 50.0 * B * (H // 2) * (W // 2) * cb * cs)
    return out


def convT_s2_dgrad_bn_supported(B, Hs, Ws, cb, cs, dtype):
    return dtype == DT_BF16 and bool(_lib.load().lg_convT_s2_dgrad_bn_supported(B, Hs, Ws, cb, cs))


def instnorm_bwd_coef(z16, stats, partials):
    """The norm backward WITHOUT its apply pass: per-sample records [B, 8] from the statistics and the producer-fused sums
    (NormPartials).  The consumer (convT_s2_dgrad_bn) forms dz while it stages its operand."""
    B = z16.shape[0]
    _chk(stats, (B, NSTAT), "stats")
    if tuple(z16.shape) != partials.shape:
        raise ValueError(f"instnorm_bwd_coef: partial sums were produced for shape {partials.shape}, got {tuple(z16.shape)}")
    coef = torch.empty(B, 8, dtype=torch.float32, device=z16.device)
    check(_lib.load().lg_instnorm_bwd_coef(_p(stats), _p(partials.buf), int(partials.nparts), _p(coef), B, z16.numel() // B, _stream()),
          "lg_instnorm_bwd_coef")
    return coef


def convT_s2_dgrad_bn(z16, g16, coef, alpha, pack, cs, fuse):
    """Data gradient of Conv2DTranspose(cb, 5, 2, same) from the level's raw pair (z16, g16) [B, 2Hs, 2Ws, cb] + coef
    (instnorm_bwd_coef): dz is formed on the fly.  fuse = (z16, stats, alpha) of the level below -> (dx16 [B, Hs, Ws, cs], NormPartials)."""
    import ctypes
    B, H, W, cb = z16.shape
    _chk16(z16, z16, "z16")
    _chk16(g16, z16, "g16")
    if tuple(g16.shape) != tuple(z16.shape):
        raise ValueError("convT_s2_dgrad_bn: z16 and g16 must have the same shape")
    _chk(coef, (B, 8), "coef")
    if not convT_s2_dgrad_bn_supported(B, H // 2, W // 2, cb, cs, DT_BF16):
        raise ValueError(f"convT_s2_dgrad_bn: unsupported shape B={B} {H // 2}x{W // 2} cb={cb} cs={cs}")
    out = torch.empty(B, H // 2, W // 2, cs, dtype=torch.bfloat16, device=z16.device)
    zl, stl, alpha_l, ws, npo = _nf_args(fuse, B, False, H // 2, W // 2, cs, z16.device)
    e0 = _pb()
    check(_lib.load().lg_convT_s2_dgrad_bn(_p(z16), _p(g16), _p(coef), float(alpha), _p(pack), _p(out), B, H // 2, W // 2, cb, cs,
                                          _p(zl), _p(stl), alpha_l, _p(ws), ws.numel(), ctypes.addressof(npo), _stream()),
          "lg_convT_s2_dgrad_bn")
    _pe(e0, "conv_igemm_down", 50.0 * B * (H // 2) * (W // 2) * cb * cs)
    if npo.value <= 0:
        raise _lib.LittleGanHipError("lg_convT_s2_dgrad_bn returned no partial sums")
    return out, NormPartials(ws, npo.value, alpha_l, out.shape)


def convT_s2_wgrad(x, dy, dw, accumulate, dtype, x16=None, dy16=None):
    """dw[5,5,cb,cs] (+)= wgrad(x [B,Hs,Ws,cs], dy [B,2Hs,2Ws,cb]); x16/dy16: optional bf16 mirrors"""
    return _wgrad("lg_convT_s2_wgrad_m16", dy, x, dw, accumulate, dtype, swap=True, big16=dy16, small16=x16)


def n3_m16_supported(H, W, cb, cs, dtype):
    """bf16 path: the 3-channel layers of this shape read the bf16 mirror of their wide operand alone."""
    return bool(_lib.load().lg_n3_m16_supported(H, W, cb, cs, dtype))


def convT_s1_tanh_fwd(x, pack, bias, cb, dtype, out=None, x16=None):
    """x16: bf16 mirror of x (bf16 path); x may be None where n3_m16_supported."""
    t = x if x is not None else x16
    B, H, W, cs = t.shape
    if x is not None:
        _chk(x, name="x")
    if x16 is not None:
        _chk16(x16, t, "x16")
    if out is None:
        out = torch.empty(B, H, W, cb, dtype=torch.float32, device=t.device)
    _chk(out, (B, H, W, cb), "out")
    _chk(bias, (cb,), "bias")
    e0 = _pb()
    check(_lib.load().lg_convT_s1_tanh_fwd_m16(_p(x), _p(x16), _p(pack), _p(bias), _p(out), B, H, W, cb, cs, dtype,
                                               _stream()), "lg_convT_s1_tanh_fwd_m16")
    _pe(e0, "conv_igemm_s1t_n3", 50.0 * B * H * W * cb * cs)
    return out


def convT_s1_tanh_fwd_z16_supported(H, W, cb, cs, dtype):
    return bool(_lib.load().lg_convT_s1_tanh_fwd_z16_supported(H, W, cb, cs, dtype))


def convT_s1_tanh_fwd_z16(z16, stats, alpha, pack, bias, cb, dtype, out=None):
    """The final layer fed with the RAW bf16 conv output z16 [B,H,W,cs] of the last decoder level and its statistics:
    InstanceNorm + LeakyReLU(alpha) happen while the kernel stages its input, the normalised map is never written."""
    B, H, W, cs = z16.shape
    _chk16(z16, z16, "z16")
    _chk(stats, (B, NSTAT), "stats")
    if out is None:
        out = torch.empty(B, H, W, cb, dtype=torch.float32, device=z16.device)
    _chk(out, (B, H, W, cb), "out")
    _chk(bias, (cb,), "bias")
    e0 = _pb()
    check(_lib.load().lg_convT_s1_tanh_fwd_z16(_p(z16), _p(stats), float(alpha), _p(pack), _p(bias), _p(out), B, H, W, cb, cs,
                                               dtype, _stream()), "lg_convT_s1_tanh_fwd_z16")
    _pe(e0, "conv_igemm_s1t_n3", 50.0 * B * H * W * cb * cs)
    return out


def convT_s1_tanh_bwd(x, dpre, pack, cs, dtype, dx=None, dw=None, db=None, accumulate=False, x16=None, dx16=None, fuse=None):
    """x16: bf16 mirror of x for the weight gradient; dx16: bf16 tensor that receives the data gradient instead of dx.
    fuse = (z16, stats, alpha) (with dx16): also the first-pass sums of the norm backward dx16 feeds -> returns
    (dx16, NormPartials or None)."""
    lib = _lib.load()
    B, H, W, cb = dpre.shape
    _chk(dpre, name="dpre")
    if x is not None:
        _chk(x, (B, H, W, cs), "x")
    if x16 is not None:
        if x16.dtype != torch.bfloat16 or tuple(x16.shape) != (B, H, W, cs) or not x16.is_contiguous():
            raise ValueError("convT_s1_tanh_bwd: x16 must be a contiguous bf16 [B,H,W,cs] tensor")
    if dx is not None:
        _chk(dx, (B, H, W, cs), "dx")
    if dx16 is not None:
        if dx16.dtype != torch.bfloat16 or tuple(dx16.shape) != (B, H, W, cs) or not dx16.is_contiguous():
            raise ValueError("convT_s1_tanh_bwd: dx16 must be a contiguous bf16 [B,H,W,cs] tensor")
    if dw is not None:
        _chk(dw, (5, 5, cb, cs), "dw")
    if db is not None:
        _chk(db, (cb,), "db")
    nbytes = int(lib.lg_convT_s1_bwd_workspace_bytes(B, H, W, cb, cs, dtype))
    ws = workspace(nbytes, dpre.device, "wgrad")
    if fuse is not None:
        import ctypes
        if dx16 is None or dx is not None:
            raise ValueError("convT_s1_tanh_bwd: fuse needs the bf16 data gradient (dx16)")
        z16, st, alpha, wsp, npo = _nf_args(fuse, B, False, H, W, cs, dpre.device)
        check(lib.lg_convT_s1_tanh_bwd_nf(_p(x), _p(x16), _p(dpre), _p(pack), _p(dx16), _p(dw), _p(db), _p(ws), ws.numel(), B, H, W,
                                          cb, cs, int(accumulate), dtype, _p(z16), _p(st), alpha, _p(wsp), wsp.numel(),
                                          ctypes.addressof(npo), _stream()), "lg_convT_s1_tanh_bwd_nf")
        return dx16, (NormPartials(wsp, npo.value, alpha, dx16.shape) if npo.value > 0 else None)
    check(lib.lg_convT_s1_tanh_bwd_m16(_p(x), _p(x16), _p(dpre), _p(pack), _p(dx), _p(dx16), _p(dw), _p(db), _p(ws),
                                       ws.numel(), B, H, W, cb, cs, int(accumulate), dtype, _stream()),
          "lg_convT_s1_tanh_bwd_m16")
    return dx if dx is not None else dx16


def bias_grad(dy, db, accumulate=False, dy16=None):
    """db (+)= column sums of dy [.., C]; dy16: bf16 copy read instead of dy (dy may then be None)"""
    t = dy if dy is not None else dy16
    C = t.shape[-1]
    M = t.numel() // C
    if dy is not None:
        _chk(dy, name="dy")
    if dy16 is not None:
        _chk16(dy16, t, "dy16")
    _chk(db, (C,), "db")
    lib = _lib.load()
    ws = workspace(int(lib.lg_bias_grad_workspace_bytes(M, C)), t.device, "small")
    check(lib.lg_bias_grad_m16(_p(dy), _p(dy16), _p(db), _p(ws), ws.numel(), M, C, int(accumulate), _stream()),
          "lg_bias_grad_m16")
    return db


_HALO_OK = {}


def conv_halo_supported(mode, dtype, B, Hm, Wm, Cs, N):
    """mode 0 = conv form ("down": M grid Hm x Wm is the SMALL map), 1 = convT form ("up").  Host-side query."""
    key = (mode, dtype, B, Hm, Wm, Cs, N)
    if key not in _HALO_OK:
        _HALO_OK[key] = bool(_lib.load().lg_conv_halo_supported(mode, dtype, B, Hm, Wm, Cs, N))
    return _HALO_OK[key]


# ------------------------------------------------------------------ instance norm
class Moments:
    """UNFINISHED InstanceNormalization moments of a conv output (bf16 activation path): the [B][nparts][3] partial records
    {count, mean, M2} the conv epilogue left in the shared 'statpart' workspace, and the [B, NSTAT] tensor the finished
    statistics go to.  The next norm call on this stream consumes them: instnorm_apply finishes them inside its own launch
    (lg_instnorm_leaky_apply_z16_p: one launch and one kernel boundary less per normalised map); any other use goes through
    stats_tensor(), which runs the stand-alone finalize.  Valid until the next conv with fused moments is enqueued."""

    def __init__(self, ws, nparts, gamma, beta, B):
        self.ws, self.nparts, self.gamma, self.beta, self.B = ws, int(nparts), gamma, beta, int(B)
        self.stats = torch.empty(B, NSTAT, dtype=torch.float32, device=ws.device)
        self.covered = 0   # rows whose records have been finished

    def __getitem__(self, rows):
        lo, hi, step = rows.indices(self.B)
        if step != 1:
            raise ValueError("Moments: contiguous row ranges only")
        return _MomentRows(self, lo, hi)


class _MomentRows:
    def __init__(self, m, lo, hi):
        self.m, self.lo, self.hi = m, lo, hi


def stats_tensor(st):
    """[B, NSTAT] statistics tensor of `st` (a tensor already, or Moments: finished here if no apply has done it)."""
    if isinstance(st, _MomentRows):
        return stats_tensor(st.m)[st.lo:st.hi]
    if not isinstance(st, Moments):
        return st
    if st.covered < st.B:
        check(_lib.load().lg_instnorm_stats_finalize(_p(st.ws), st.nparts, _p(st.stats), _p(st.gamma), _p(st.beta), st.B, _stream()),
              "lg_instnorm_stats_finalize")
        st.covered = st.B
    return st.stats


def instnorm_stats(x, gamma, beta, pre_leaky, alpha, stats=None, x16_out=None):
    """x16_out (optional bf16 tensor like x): also receives bf16(x) in the same pass."""
    B = x.shape[0]
    Ln = x.numel() // B
    _chk(x, name="x")
    if stats is None:
        stats = torch.empty(B, NSTAT, dtype=torch.float32, device=x.device)
    _chk(stats, (B, NSTAT), "stats")
    lib = _lib.load()
    ws = workspace(int(lib.lg_instnorm_workspace_bytes(B, Ln)), x.device, "small")
    if x16_out is not None:
        _chk16(x16_out, x, "x16_out")
    check(lib.lg_instnorm_leaky_stats_z16(_p(x), _p(stats), _p(gamma), _p(beta), _p(ws), ws.numel(), B, Ln, int(pre_leaky),
                                          float(alpha), _p(x16_out), _stream()), "lg_instnorm_leaky_stats")
    return stats


def _chk16(t, like, name):
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() == like.numel()):
        raise ValueError(f"{name}: need a contiguous bf16 CUDA tensor with {like.numel()} elements")
    return t


def instnorm_apply(x, stats, skip, pre_leaky, post_leaky, alpha, out=None, out16=None, want_f32=True):
    """want_f32=False: only the bf16 mirror out16 is written (returns None).
    x may be the bf16 conv output of the bf16 activation path (then skip may be bf16 too)."""
    B = x.shape[0]
    Ln = x.numel() // B
    x_is16 = x.dtype == torch.bfloat16
    if x_is16:
        _chk16(x, x, "x")
        if Ln % 8:
            raise ValueError("instnorm_apply: a bf16 input needs a multiple of 8 elements per sample")
    else:
        _chk(x, name="x")
    pend = None   # unfinished moments (Moments / a row range of them): finished inside the apply launch where the kernel exists
    if isinstance(stats, (Moments, _MomentRows)):
        mrows = stats if isinstance(stats, _MomentRows) else _MomentRows(stats, 0, stats.B)
        if mrows.hi - mrows.lo != B:
            raise ValueError("instnorm_apply: moments cover a different number of samples")
        if x_is16 and not pre_leaky and mrows.m.covered < mrows.m.B:
            pend = mrows
            stats = mrows.m.stats[mrows.lo:mrows.hi]
        else:
            stats = stats_tensor(stats)
    _chk(stats, (B, NSTAT), "stats")
    skip16 = skip is not None and skip.dtype == torch.bfloat16
    if skip is not None:
        if skip16:
            if not x_is16:
                raise ValueError("instnorm_apply: a bf16 skip needs the bf16 input path")
            _chk16(skip, x, "skip")
        else:
            _chk(skip, name="skip")
        if skip.numel() != x.numel():
            raise ValueError("instnorm_apply: skip has a different size")
    if not want_f32:
        if out16 is None:
            raise ValueError("instnorm_apply: want_f32=False needs out16")
        out = None
    else:
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        _chk(out, x.shape, "out")
    if out16 is not None:
        _chk16(out16, x, "out16")
    if pend is not None:
        m = pend.m
        part = m.ws.data_ptr() + pend.lo * m.nparts * 3 * 8
        check(_lib.load().lg_instnorm_leaky_apply_z16_p(_p(x), part, m.nparts, _p(m.gamma), _p(m.beta), _p(stats), _p(skip), int(skip16),
                                                        _p(out), _p(out16), B, Ln, int(post_leaky), float(alpha), _stream()),
              "lg_instnorm_leaky_apply_z16_p")
        m.covered += B
    elif x_is16:
        check(_lib.load().lg_instnorm_leaky_apply_z16(_p(x), _p(stats), _p(skip), int(skip16), _p(out), _p(out16), B, Ln,
                                                      int(pre_leaky), int(post_leaky), float(alpha), _stream()),
              "lg_instnorm_leaky_apply_z16")
    else:
        check(_lib.load().lg_instnorm_leaky_apply(_p(x), _p(stats), _p(skip), _p(out), _p(out16), B, Ln, int(pre_leaky),
                                                  int(post_leaky), float(alpha), _stream()), "lg_instnorm_leaky_apply")
    return out


def instnorm_bwd(x, stats, g, dgamma, dbeta, pre_leaky, post_leaky, alpha, accumulate=False, out=None, out16=None,
                 want_f32=True, db=None, partials=None):
    """g may be fp32 or bf16 (as written by a bf16 data-gradient conv); x fp32, or the bf16 conv output of the bf16
    activation path.  Returns the fp32 dx (or None if want_f32 is False, in which case only the bf16 mirror out16 is
    written).  db [C] (optional, C = x.shape[-1]): receives the column sums of dx = the bias gradient of the conv
    layer that produced x, in the same pass."""
    B = x.shape[0]
    Ln = x.numel() // B
    x_is16 = x.dtype == torch.bfloat16
    if x_is16:
        _chk16(x, x, "x")
        if Ln % 8:
            raise ValueError("instnorm_bwd: a bf16 input needs a multiple of 8 elements per sample")
    else:
        _chk(x, name="x")
    g16 = g.dtype == torch.bfloat16
    if g16:
        _chk16(g, x, "g")
    else:
        _chk(g, name="g")
        if g.numel() != x.numel():
            raise ValueError("instnorm_bwd: g has a different size")
    _chk(stats, (B, NSTAT), "stats")
    if want_f32:
        if out is None:
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        _chk(out, x.shape, "out")
    else:
        if out16 is None:
            raise ValueError("instnorm_bwd: want_f32=False needs out16")
        out = None
    if out16 is not None:
        _chk16(out16, x, "out16")
    lib = _lib.load()
    C = 0
    if db is not None:
        C = x.shape[-1]
        _chk(db, (C,), "db")
    ws = workspace(int(lib.lg_instnorm_bwd_db_workspace_bytes(B, Ln, C)), x.device, "small")
    if partials is not None:  # NormPartials from the conv that produced g: the first pass over (x, g) is skipped
        if not x_is16:
            raise ValueError("instnorm_bwd: fused partial sums belong to the bf16 activation path")
        # the producer's store loop hard-codes g' = LeakyReLU'_alpha(a c + b) g (lg_nf_accum): anything else is another sum
        if pre_leaky or not post_leaky or float(alpha) != partials.alpha or tuple(x.shape) != partials.shape:
            raise ValueError(f"instnorm_bwd: partial sums were produced for the post-LeakyReLU(alpha={partials.alpha}) form on "
                             f"shape {partials.shape}; asked for pre={pre_leaky} post={post_leaky} alpha={alpha} shape {tuple(x.shape)}")
        check(lib.lg_instnorm_leaky_bwd_z16_p(_p(x), _p(stats), _p(g), int(g16), _p(out), _p(out16), _p(dgamma), _p(dbeta), _p(db), C,
                                              _p(partials.buf), int(partials.nparts), _p(ws), ws.numel(), B, Ln, int(pre_leaky),
                                              int(post_leaky), float(alpha), int(accumulate), _stream()),
              "lg_instnorm_leaky_bwd_z16_p")
        return out
    fn = lib.lg_instnorm_leaky_bwd_z16 if x_is16 else lib.lg_instnorm_leaky_bwd_db
    check(fn(_p(x), _p(stats), _p(g), int(g16), _p(out), _p(out16), _p(dgamma), _p(dbeta), _p(db), C, _p(ws), ws.numel(), B, Ln,
             int(pre_leaky), int(post_leaky), float(alpha), int(accumulate), _stream()), "lg_instnorm_leaky_bwd")
    return out


# ------------------------------------------------------------------ dense
def dense_fwd(x, w, bias, out=None):
    B, K = x.shape
    N = w.shape[1]
    _chk(x, name="x")
    _chk(w, (K, N), "w")
    if out is None:
        out = torch.empty(B, N, dtype=torch.float32, device=x.device)
    _chk(out, (B, N), "out")
    check(_lib.load().lg_dense_fwd(_p(x), _p(w), _p(bias), _p(out), B, K, N, _stream()), "lg_dense_fwd")
    return out


def dense_wgrad(x, dy, dw, db, accumulate=False):
    B, K = x.shape
    N = dy.shape[1]
    _chk(x, name="x")
    _chk(dy, (B, N), "dy")
    _chk(dw, (K, N), "dw")
    check(_lib.load().lg_dense_wgrad(_p(x), _p(dy), _p(dw), _p(db), B, K, N, int(accumulate), _stream()),
          "lg_dense_wgrad")


def concat_cols(a, c, out=None):
    """[a | c] along the last axis (the Generator's dense input, model.py:97-98)."""
    B, ka = a.shape
    kc = c.shape[1]
    _chk(a, name="a")
    _chk(c, (B, kc), "c")
    if out is None:
        out = torch.empty(B, ka + kc, dtype=torch.float32, device=a.device)
    _chk(out, (B, ka + kc), "out")
    check(_lib.load().lg_concat_cols(_p(a), ka, _p(c), kc, _p(out), B, _stream()), "lg_concat_cols")
    return out


def adj_conditions(first, second):
    """(t, u) = (concat([first, second], 0), (t + 1) * 0.5): the Adjuster's target / input conditions (eager_trainer.py:153-154)."""
    B, c = first.shape
    _chk(first, name="first")
    _chk(second, (B, c), "second")
    t = torch.empty(2 * B, c, dtype=torch.float32, device=first.device)
    u = torch.empty_like(t)
    check(_lib.load().lg_adj_conditions(_p(first), _p(second), _p(t), _p(u), B, c, _stream()), "lg_adj_conditions")
    return t, u


def dense_dgrad(dy, w, out=None):
    B, N = dy.shape
    K = w.shape[0]
    _chk(dy, name="dy")
    _chk(w, (K, N), "w")
    if out is None:
        out = torch.empty(B, K, dtype=torch.float32, device=dy.device)
    _chk(out, (B, K), "out")
    check(_lib.load().lg_dense_dgrad(_p(dy), _p(w), _p(out), B, K, N, _stream()), "lg_dense_dgrad")
    return out


def heads_fwd(x, wpr, bpr, wc, bc, out=None):
    B, K = x.shape
    c = wc.shape[1]
    _chk(x, name="x")
    _chk(wpr, (K, 1), "wpr")
    _chk(wc, (K, c), "wc")
    if out is None:
        out = torch.empty(B, 1 + c, dtype=torch.float32, device=x.device)
    _chk(out, (B, 1 + c), "out")
    lib = _lib.load()
    ws = workspace(int(lib.lg_heads_fwd_workspace_bytes(B, K, c)), x.device, "small")
    check(lib.lg_heads_fwd(_p(x), _p(wpr), _p(bpr), _p(wc), _p(bc), _p(out), _p(ws), ws.numel(), B, K, c, _stream()),
          "lg_heads_fwd")
    return out


def heads_dgrad(dz, wpr, wc, out=None):
    B = dz.shape[0]
    K, c = wc.shape
    _chk(dz, (B, 1 + c), "dz")
    if out is None:
        out = torch.empty(B, K, dtype=torch.float32, device=dz.device)
    _chk(out, (B, K), "out")
    check(_lib.load().lg_heads_dgrad(_p(dz), _p(wpr), _p(wc), _p(out), B, K, c, _stream()), "lg_heads_dgrad")
    return out


def heads_wgrad(x, dz, dwpr, dbpr, dwc, dbc, accumulate=False):
    B, K = x.shape
    c = dwc.shape[1]
    _chk(x, name="x")
    _chk(dz, (B, 1 + c), "dz")
    _chk(dwpr, (K, 1), "dwpr")
    _chk(dwc, (K, c), "dwc")
    check(_lib.load().lg_heads_wgrad(_p(x), _p(dz), _p(dwpr), _p(dbpr), _p(dwc), _p(dbc), B, K, c, int(accumulate),
                                     _stream()), "lg_heads_wgrad")


# ------------------------------------------------------------------ losses / optimizer
def bce_heads_loss(p, t_c, t_pr, w_pr, w_c, loss, dz, accumulate):
    B, J = p.shape
    _chk(p, name="p")
    _chk(dz, (B, J), "dz")
    if t_c is not None:
        _chk(t_c, (B, J - 1), "t_c")
    check(_lib.load().lg_bce_heads_loss_fwd_bwd(_p(p), _p(t_c), float(t_pr), float(w_pr), float(w_c), _p(loss), _p(dz),
                                                B, J - 1, int(accumulate), _stream()), "lg_bce_heads_loss_fwd_bwd")


def l1_tanh_loss(t, img, g_in, dpre, loss, lam, accumulate):
    _chk(t, name="t")
    _chk(img, t.shape, "img")
    if g_in is not None:
        _chk(g_in, t.shape, "g_in")
    if dpre is not None:
        _chk(dpre, t.shape, "dpre")
    lib = _lib.load()
    ws = workspace(int(lib.lg_l1_workspace_bytes()), t.device, "small")
    check(lib.lg_l1_tanh_loss_fwd_bwd(_p(t), _p(img), _p(g_in), _p(dpre), _p(loss), _p(ws), ws.numel(), t.numel(),
                                      float(lam), int(accumulate), _stream()), "lg_l1_tanh_loss_fwd_bwd")


def clip_adam_update(w, g, m, v, state, lr, b1, b2, eps, clip, gscale=1.0):
    n = w.numel()
    for t, nm in ((w, "w"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, name=nm)
        if t.numel() != n:
            raise ValueError("clip_adam_update: size mismatch")
    check(_lib.load().lg_clip_adam_update(_p(w), _p(g), _p(m), _p(v), n, _p(state), float(lr), float(b1), float(b2),
                                          float(eps), float(clip), float(gscale), _stream()), "lg_clip_adam_update")


def adam_advance(state, b1, b2):
    check(_lib.load().lg_adam_advance(_p(state), float(b1), float(b2), _stream()), "lg_adam_advance")


# ------------------------------------------------------------------ conv forward with fused InstanceNorm moments
def _fwd_stats(fn_name, x, x16, pack, bias, out, B, Hs, Ws, cb, cs, dtype, gamma, beta, tag, flops, up, defer=False):
    """Runs the conv (`out` fp32, or bf16 = the bf16 activation path); if its kernel produced per-block moment partials,
    finishes them into the stats record.  Returns stats [B, NSTAT] or None (caller then runs instnorm_stats)."""
    import ctypes
    lib = _lib.load()
    ws = workspace(int(lib.lg_conv_stats_workspace_bytes(int(up), B, Hs, Ws, cb if up else cs)), out.device, "statpart")
    nparts = ctypes.c_int(0)
    o16 = out.dtype == torch.bfloat16
    e0 = _pb()
    if x16 is not None:
        _chk16(x16, x if x is not None else x16, "x16")
    check(getattr(lib, fn_name)(_p(x), _p(x16), _p(pack), _p(bias), 0 if o16 else _p(out), _p(out) if o16 else 0, B, Hs, Ws, cb,
                                cs, dtype, _p(ws), ws.numel(), ctypes.addressof(nparts), _stream()), fn_name)
    _pe(e0, tag, flops)
    if nparts.value <= 0:
        return None
    if defer:   # the consumer (instnorm_apply) finishes the records in its own launch
        return Moments(ws, nparts.value, gamma, beta, B)
    stats = torch.empty(B, NSTAT, dtype=torch.float32, device=out.device)
    check(lib.lg_instnorm_stats_finalize(_p(ws), nparts.value, _p(stats), _p(gamma), _p(beta), B, _stream()),
          "lg_instnorm_stats_finalize")
    return stats


_FUSED_OK = {}


def _fwd_stats_z16(fn_name, x, x16, pack, bias, shape, B, Hs, Ws, cb, cs, dtype, gamma, beta, tag, flops, up, alpha, defer=False):
    """bf16 activation path: z leaves the conv as bf16.  Where the conv kernel fuses the moments (from its fp32
    accumulators) that is the only copy ever written; otherwise the conv writes fp32 once, the statistics pass reads it
    and emits the bf16 copy in the same sweep, and the fp32 tensor is dropped.  Either way:
    moments of the fp32 z, everything downstream reads bf16(z)."""
    dev = bias.device
    key = (int(up), dtype, B, Hs, Ws, cb, cs)
    if key not in _FUSED_OK:
        _FUSED_OK[key] = bool(_lib.load().lg_conv_fwd_stats_fused(*key))
    if _FUSED_OK[key]:
        z16 = torch.empty(shape, dtype=torch.bfloat16, device=dev)
        st = _fwd_stats(fn_name, x, x16, pack, bias, z16, B, Hs, Ws, cb, cs, dtype, gamma, beta, tag, flops, up, defer=defer)
        if st is None:
            raise _lib.LittleGanHipError(f"{fn_name}: lg_conv_fwd_stats_fused promised fused moments for {key}, none came")
        return z16, st
    z = torch.empty(shape, dtype=torch.float32, device=dev)
    st = _fwd_stats(fn_name, x, x16, pack, bias, z, B, Hs, Ws, cb, cs, dtype, gamma, beta, tag, flops, up)
    if st is not None:  # (moments fused although not promised: keep them, cast once)
        return z.to(torch.bfloat16), st
    z16 = torch.empty(shape, dtype=torch.bfloat16, device=dev)
    st = instnorm_stats(z, gamma, beta, 0, alpha, x16_out=z16)
    return z16, st


def conv2d_s2_fwd_stats(x, pack, bias, cs, dtype, gamma, beta, x16=None, z16=False, alpha=0.3, defer_stats=False):
    """conv2d_s2_fwd + the InstanceNormalization statistics of its output -> (z, stats or None).
    z16=True (bf16 dtype): z is returned as a bf16 tensor and stats is never None (see _fwd_stats_z16).
    defer_stats (with z16): where the conv fuses the moments, `stats` comes back as an unfinished ops.Moments for the next
    instnorm_apply to finish in its own launch (ops.stats_tensor() for any other use)."""
    B, H, W, cb = (x if x is not None else x16).shape  # x may be None when the bf16 mirror feeds the halo kernel
    if x is not None:
        _chk(x, name="x")
    _chk(bias, (cs,), "bias")
    if H % 2 or W % 2:
        raise ValueError("conv2d_s2_fwd_stats: H and W must be even")
    tag, fl = "conv_igemm_patch" if cb == 3 else "conv_igemm_down", 50.0 * B * (H // 2) * (W // 2) * cb * cs
    if z16:
        return _fwd_stats_z16("lg_conv2d_s2_fwd_stats", x, x16, pack, bias, (B, H // 2, W // 2, cs), B, H // 2, W // 2, cb, cs,
                              dtype, gamma, beta, tag, fl, False, alpha, defer=defer_stats)
    out = torch.empty(B, H // 2, W // 2, cs, dtype=torch.float32, device=bias.device)
    st = _fwd_stats("lg_conv2d_s2_fwd_stats", x, x16, pack, bias, out, B, H // 2, W // 2, cb, cs, dtype, gamma, beta, tag, fl,
                    up=False)
    return out, st


_ZN_OK = {}


def conv2d_s2_fwd_stats_zn_supported(B, H, W, cb, cs, dtype):
    """H x W: the map the conv reads (as conv2d_s2_fwd_stats).  True where conv2d_s2_fwd_stats_zn runs."""
    key = (B, H // 2, W // 2, cb, cs, dtype)
    if key not in _ZN_OK:
        _ZN_OK[key] = H % 2 == 0 and W % 2 == 0 and bool(_lib.load().lg_conv2d_s2_fwd_stats_zn_supported(*key))
    return _ZN_OK[key]


def conv2d_s2_fwd_stats_zn(zin16, zin_stats, alpha, pack, bias, cs, dtype, gamma, beta):
    """conv2d_s2_fwd_stats (bf16 path) fed with the RAW bf16 output zin16 [B,H,W,cb] of the level below and its statistics
    records: InstanceNorm + LeakyReLU(alpha) happen while the conv stages its operand, the normalised map is never written
    (bit-identical to instnorm_apply + conv2d_s2_fwd_stats).  -> (z16 [B,H/2,W/2,cs] bf16, stats [B, NSTAT])."""
    import ctypes
    lib = _lib.load()
    B, H, W, cb = zin16.shape
    _chk16(zin16, zin16, "zin16")
    _chk(zin_stats, (B, NSTAT), "zin_stats")
    _chk(bias, (cs,), "bias")
    if not conv2d_s2_fwd_stats_zn_supported(B, H, W, cb, cs, dtype):
        raise ValueError(f"conv2d_s2_fwd_stats_zn: shape {tuple(zin16.shape)} -> {cs} is outside the normalising kernel's tiling")
    Hs, Ws = H // 2, W // 2
    ws = workspace(int(lib.lg_conv_stats_workspace_bytes(0, B, Hs, Ws, cs)), zin16.device, "statpart")
    z16 = torch.empty(B, Hs, Ws, cs, dtype=torch.bfloat16, device=zin16.device)
    nparts = ctypes.c_int(0)
    e0 = _pb()
    check(lib.lg_conv2d_s2_fwd_stats_zn(_p(zin16), _p(zin_stats), float(alpha), _p(pack), _p(bias), _p(z16), B, Hs, Ws, cb, cs, dtype,
                                        _p(ws), ws.numel(), ctypes.addressof(nparts), _stream()), "lg_conv2d_s2_fwd_stats_zn")
    _pe(e0, "conv_igemm_down", 50.0 * B * Hs * Ws * cb * cs)
    if nparts.value <= 0:
        raise _lib.LittleGanHipError("lg_conv2d_s2_fwd_stats_zn: no moment partials came back")
    stats = torch.empty(B, NSTAT, dtype=torch.float32, device=zin16.device)
    check(lib.lg_instnorm_stats_finalize(_p(ws), nparts.value, _p(stats), _p(gamma), _p(beta), B, _stream()),
          "lg_instnorm_stats_finalize")
    return z16, stats


def convT_s2_fwd_stats(x, pack, bias, cb, dtype, gamma, beta, x16=None, z16=False, alpha=0.3, defer_stats=False):
    B, Hs, Ws, cs = (x if x is not None else x16).shape
    if x is not None:
        _chk(x, name="x")
    _chk(bias, (cb,), "bias")
    fl = 50.0 * B * Hs * Ws * cb * cs
    if z16:
        return _fwd_stats_z16("lg_convT_s2_fwd_stats", x, x16, pack, bias, (B, 2 * Hs, 2 * Ws, cb), B, Hs, Ws, cb, cs, dtype,
                              gamma, beta, "conv_igemm_up", fl, True, alpha, defer=defer_stats)
    out = torch.empty(B, 2 * Hs, 2 * Ws, cb, dtype=torch.float32, device=bias.device)
    st = _fwd_stats("lg_convT_s2_fwd_stats", x, x16, pack, bias, out, B, Hs, Ws, cb, cs, dtype, gamma, beta, "conv_igemm_up", fl,
                    up=True)
    return out, st


# ------------------------------------------------------------------ step inputs on the device (eager_trainer.py:125-131)
def _i64(v):
    """unsigned 64-bit value as the signed pattern ctypes' c_longlong takes"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def philox4x32(nblocks, seed, offset, device="cuda"):
    out = torch.empty(nblocks * 4, dtype=torch.int32, device=device)
    check(_lib.load().lg_philox4x32(_p(out), nblocks, _i64(seed), _i64(offset), _stream()), "lg_philox4x32")
    return out


def randn(shape, seed, offset, mean=0.0, std=1.0, device="cuda"):
    """tf.random.normal(shape, mean, std) from the counter-based generator: reproducible per (seed, offset)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(_lib.load().lg_randn(_p(out), out.numel(), float(mean), float(std), _i64(seed), _i64(offset), _stream()), "lg_randn")
    return out


def augment(img, flip, db, cf, dh, noise_scale, seed, offset, out=None):
    """flip: uint8 [B] device tensor or None; see lg_augment."""
    B, H, W, c = img.shape
    _chk(img, name="img")
    if c != 3:
        raise ValueError("augment: 3-channel images only")
    if out is None:
        out = torch.empty_like(img)
    _chk(out, img.shape, "out")
    if flip is not None and not (flip.is_cuda and flip.dtype == torch.uint8 and flip.numel() == B and flip.is_contiguous()):
        raise ValueError("augment: flip must be a contiguous uint8 CUDA tensor with B elements")
    lib = _lib.load()
    ws = workspace(int(lib.lg_augment_workspace_bytes(B)), img.device, "small")
    check(lib.lg_augment(_p(img), _p(out), B, H, W, _p(flip), float(db), float(cf), float(dh), float(noise_scale),
                         _i64(seed), _i64(offset), _p(ws), ws.numel(), _stream()), "lg_augment")
    return out


def augment_drawn(img, db_max, c_lo, c_hi, dh_max, noise_scale, seed, draw_offset, noise_offset, out=None):
    """lg_augment_drawn: flip / brightness / contrast / hue draws made on the device from the Philox window at
    draw_offset (no host synchronisation); see include/littlegan_hip.h."""
    B, H, W, c = img.shape
    _chk(img, name="img")
    if c != 3:
        raise ValueError("augment_drawn: 3-channel images only")
    if out is None:
        out = torch.empty_like(img)
    _chk(out, img.shape, "out")
    lib = _lib.load()
    ws = workspace(int(lib.lg_augment_drawn_workspace_bytes(B)), img.device, "small")
    check(lib.lg_augment_drawn(_p(img), _p(out), B, H, W, float(db_max), float(c_lo), float(c_hi), float(dh_max),
                               float(noise_scale), _i64(seed), _i64(draw_offset), _i64(noise_offset), _p(ws), ws.numel(),
                               _stream()), "lg_augment_drawn")
    return out


def fid_stats(act):
    """fid.py:185-188 on the device: (mu [D], sigma [D, D]) fp64 tensors of act [N, D] (fp32 CUDA)."""
    if act.dim() != 2 or act.shape[0] < 2:
        raise ValueError("fid_stats: need an [N >= 2, D] matrix")
    _chk(act, name="act")
    N, D = act.shape
    mu = torch.empty(D, dtype=torch.float64, device=act.device)
    sigma = torch.empty(D, D, dtype=torch.float64, device=act.device)
    lib = _lib.load()
    ws = workspace(int(lib.lg_fid_stats_workspace_bytes(N, D)), act.device, "small")
    check(lib.lg_fid_stats(_p(act), N, D, _p(mu), _p(sigma), _p(ws), ws.numel(), _stream()), "lg_fid_stats")
    return mu, sigma
