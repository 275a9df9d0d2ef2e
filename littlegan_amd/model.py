"""Host-side mirror of /root/reference/model.py on the HIP kernels (no autograd, no fallback).

Same constructors and call signatures as the reference (main.py:20-24):
    Decoder(args), Encoder(args), Generator(args, decoder), Discriminator(args, encoder),
    Adjuster(args, discriminator, generator)
    G([noise, cond]) -> image[B,H,W,3];  D(image) -> (pr[B,1], c[B,cond_dim]);  A([image, cond]) -> image
and the same `.weights` ordering (eager_trainer.py:48-63 indexes into it): Generator 22,
Discriminator 20, Adjuster 38 (own = [16:20]).  Tensors are NHWC fp32 torch CUDA tensors.

Every forward can record a context (`ctx`) holding exactly what the hand-written backward needs:
the layer inputs, the raw (pre-norm) conv outputs and the per-sample norm statistics.  The
backward methods compute the gradient sets of eager_trainer.py:145,149,163 and nothing else.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import DT_BF16, DT_F32


import os as _os
_ZN = not _os.environ.get("LG_NO_ZN")   # A/B switch: apply passes left to the consuming conv where no one else reads the map
_BN = not _os.environ.get("LG_NO_BWDNORM")   # A/B switch: norm-backward apply left to the consuming data-gradient conv (no-weight-gradient levels)
_DEFER = not _os.environ.get("LG_NO_DEFER")   # A/B switch: moments finished by the apply launch (default) or by their own kernel


def _dtype_of(args) -> int:
    name = getattr(args, "mfma_dtype", "f32")
    if name not in ("f32", "bf16"):
        raise ValueError(f"mfma_dtype must be 'f32' or 'bf16', got {name!r}")
    return DT_BF16 if name == "bf16" else DT_F32


def _device_of(args):
    return torch.device(getattr(args, "device", "cuda"))


def _glorot(shape, gen, device):
    """TF default kernel initializer (glorot_uniform) of tf.compat.v1.layers.{Dense,Conv2D,Conv2DTranspose}."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = math.prod(shape[:-2])
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * lim).to(device)


class _Module:
    """Minimal stand-in for keras.Model: ordered named weights + optional flat-store gradients."""

    def __init__(self, args):
        self.args = args
        self.device = _device_of(args)
        self.dtype = _dtype_of(args)
        self._names: List[str] = []
        self._w: Dict[str, torch.Tensor] = {}
        self._g: Dict[str, torch.Tensor] = {}  # gradient views, installed by ParamStore.adopt
        seed = getattr(args, "seed", 0)
        self._gen = torch.Generator().manual_seed(int(seed) * 1000 + zlib.crc32(type(self).__name__.encode()) % 1000)

    def _add(self, name, shape, kind):
        if kind == "kernel":
            t = _glorot(tuple(shape), self._gen, self.device)
        elif kind == "ones":
            t = torch.ones(shape, dtype=torch.float32, device=self.device)
        else:
            t = torch.zeros(shape, dtype=torch.float32, device=self.device)
        self._names.append(name)
        self._w[name] = t

    @property
    def weights(self) -> List[torch.Tensor]:
        return [self._w[n] for n in self._names]

    def own_items(self):
        return [(n, self._w[n]) for n in self._names]

    def grad(self, name) -> Optional[torch.Tensor]:
        return self._g.get(name)


class _ConvStack(_Module):
    """Four 5x5 stride-2 layers + InstanceNormalization, shared machinery of Encoder / Decoder."""

    def __init__(self, args, chans):
        super().__init__(args)
        self.chans = chans  # [(cb, cs)] per layer
        self._packs: List[Optional[torch.Tensor]] = [None] * 4
        self._pack_version = -1
        self.version = 0  # bumped by whoever updates the weights
        for i, (cb, cs) in enumerate(chans, 1):
            self._add(f"conv{i}.kernel", (5, 5, cb, cs), "kernel")
            self._add(f"conv{i}.bias", (self._bias_dim(cb, cs),), "zeros")
            self._add(f"norm{i}.gamma", (1,), "ones")
            self._add(f"norm{i}.beta", (1,), "zeros")

    def packs(self):
        if self._pack_version != self.version:
            for i, (cb, cs) in enumerate(self.chans):
                self._packs[i] = ops.conv_pack(self._w[f"conv{i + 1}.kernel"], cb, cs, self.dtype, out=self._packs[i])
            self._pack_version = self.version
        return self._packs


class Encoder(_ConvStack):
    """model.py:6-27.  conv_i: Conv2D(conv_filter[4-i], 5, 2, 'same') -> InstanceNorm -> leaky -> dropout(identity)."""

    def __init__(self, args):
        cf = args.conv_filter
        chans, cin = [], args.image_channel
        for i in range(1, 5):
            chans.append((cin, cf[4 - i]))
            cin = cf[4 - i]
        super().__init__(args, chans)

    @staticmethod
    def _bias_dim(cb, cs):
        return cs

    def __call__(self, inputs, ctx: Optional[dict] = None, tails=None, keep_maps: bool = True, top_only: bool = False):
        """Returns the 4 maps (model.py:27).  f32 path: fp32 tensors.  bf16 path: maps 1-3 are the bf16 mirrors the next
        conv reads anyway (they are also what the Adjuster's decoder adds as skips), map 4 (8x8, heads input) is fp32;
        the raw conv outputs z are kept in HBM as bf16 only (moments from the fp32 accumulators of the conv epilogue).
        keep_maps=False (f32 path): the caller only uses the LAST map.
        tails (optional): the 4 maps this encoder already produced for MORE samples that follow `inputs` in the
        batch (same weights); each returned map is then the pair (own map, tail) instead of a recomputation or a
        concatenated copy (the Adjuster's input is [img1 ; fake] and D has just encoded `fake`).  Every op is
        per-sample, so the result is identical.
        top_only (bf16 path): the caller reads the LAST map only and no tape will ask this pass for a weight gradient (D on the
        Adjuster's output, eager_trainer.py:158-160) — the normalised maps 1-3 then have one reader, the next conv, and where its
        kernel can normalise while it stages (ops.conv2d_s2_fwd_stats_zn) they are never written: outs[i] is None there and the
        context holds no x / x16 for the level above.  Bit-identical results."""
        x = inputs
        a = self.args.leaky_alpha
        packs = self.packs()
        outs = []
        saved = []
        m16 = self.dtype == DT_BF16  # bf16 MFMA path: keep a bf16 mirror of every conv input (the operand image)
        x16 = None
        if top_only and (tails is not None or not m16):
            top_only = False
        raw = None   # (z, stats) of the level below when its apply pass was left out
        for i, (cb, cs) in enumerate(self.chans, 1):
            gm, bt = self._w[f"norm{i}.gamma"], self._w[f"norm{i}.beta"]
            # does the NEXT conv normalise this level's output itself?
            Bn, Hn, Wn = (x if x is not None else x16 if x16 is not None else raw[0]).shape[:3]
            skip_apply = (top_only and _ZN and i < 4 and
                          ops.conv2d_s2_fwd_stats_zn_supported(Bn, Hn // 2, Wn // 2, cs, self.chans[i][1], self.dtype))
            if raw is not None:
                z, st = ops.conv2d_s2_fwd_stats_zn(raw[0], raw[1], a, packs[i - 1], self._w[f"conv{i}.bias"], cs, self.dtype, gm, bt)
            else:
                z, st = ops.conv2d_s2_fwd_stats(x, packs[i - 1], self._w[f"conv{i}.bias"], cs, self.dtype, gm, bt, x16=x16,
                                                z16=m16, alpha=a, defer_stats=m16 and _DEFER and not skip_apply)
            if st is None:  # kernel without the fused-moments epilogue (small maps, 3-channel input)
                st = ops.instnorm_stats(z, gm, bt, 0, a)
            if skip_apply:
                st = ops.stats_tensor(st)
                outs.append(None)
                saved.append((x, z, st, x16))
                x, x16, raw = None, None, (z, st)
                continue
            if m16:
                h16 = torch.empty(z.shape, dtype=torch.bfloat16, device=z.device)
                # an fp32 copy only where something other than an MFMA operand load reads it: the top map (heads, first
                # skip add) and inputs of shapes only the per-tap gather kernel covers
                need32 = i == 4 or not ops.conv_halo_supported(0, self.dtype, z.shape[0], z.shape[1] // 2, z.shape[2] // 2, cs,
                                                               self.chans[i][1])
                h = ops.instnorm_apply(z, st, None, 0, 1, a, out16=h16, want_f32=need32)   # (finishes deferred moments itself)
                st = ops.stats_tensor(st)
                m = h if i == 4 else h16
            else:
                h16 = None
                h = m = ops.instnorm_apply(z, st, None, 0, 1, a)
            outs.append(m if tails is None else (m, tails[i - 1]))  # pair (own part, tail): no concatenated copy is made
            saved.append((x, z, st, x16))
            x, x16, raw = h, h16, None
        if ctx is not None:
            ctx["enc"] = saved
            ctx["enc_maps"] = outs
        return outs

    def backward(self, ctx, g_last, need_wgrad: bool, need_input_grad: bool, rows: Optional[slice] = None,
                 wgrad_levels=None):
        """g_last: gradient w.r.t. the LAST returned map (the only one any tape of the step uses).
        rows: restrict to a batch slice of the recorded context (the fake half of a [real;fake] batch).
        wgrad_levels (with need_wgrad): the levels (1..4) whose 4 weights are differentiated — a partition step
        (eager_trainer.py:104-113) trains one weight group only, and like the reference's tape.gradient the chain then
        stops at the lowest level anything is asked of."""
        a = self.args.leaky_alpha
        packs = self.packs()
        g_h = g_last
        levels = set(range(1, 5)) if (need_wgrad and wgrad_levels is None) else (set(wgrad_levels or ()) if need_wgrad else set())
        lowest = 1 if need_input_grad else (min(levels) if levels else 5)
        any_wgrad = need_wgrad
        nfp = None  # first-pass sums of this level's norm backward, if the conv that wrote g_h produced them
        for i in range(4, lowest - 1, -1):
            need_wgrad = any_wgrad and i in levels
            cb, cs = self.chans[i - 1]
            x, z, st, x16 = ctx["enc"][i - 1]
            if need_wgrad and x is None and x16 is None:
                raise ValueError("Encoder.backward: this context was recorded with top_only=True (no normalised inputs kept): "
                                 "it cannot give weight gradients")
            if rows is not None:
                z, st = z[rows], st[rows]
                x = x[rows] if x is not None else None
                x16 = x16[rows] if x16 is not None else None
            dgm = self._g[f"norm{i}.gamma"] if need_wgrad else None
            dbt = self._g[f"norm{i}.beta"] if need_wgrad else None
            want_dx = i > lowest or (i == 1 and need_input_grad)
            m16 = self.dtype == DT_BF16
            # the fp32 copy of dz is dead when every consumer reads the bf16 mirror: the data-gradient conv (halo kernel,
            # or the tap-product kernel of the 3-channel level 1), the weight-gradient kernel (needs the mirror of x too;
            # level 1: reads the fp32 image and the mirror of dz) and the bias column sums
            if cb != 3:
                dz16 = torch.empty(z.shape, dtype=torch.bfloat16, device=z.device) if m16 else None
                drop32 = (m16 and (not need_wgrad or x16 is not None) and
                          (not want_dx or ops.conv_halo_supported(1, self.dtype, z.shape[0], z.shape[1], z.shape[2], cs, cb)))
            else:
                drop32 = m16 and ops.n3_m16_supported(z.shape[1], z.shape[2], cb, cs, self.dtype)
                dz16 = torch.empty(z.shape, dtype=torch.bfloat16, device=z.device) if drop32 else None
            dz = ops.instnorm_bwd(z, st, g_h, dgm, dbt, 0, 1, a, out16=dz16, want_f32=not drop32,
                                  db=self._g[f"conv{i}.bias"] if need_wgrad else None,  # bias gradient = column sums of dz
                                  partials=nfp)
            nfp = None
            if need_wgrad:
                ops.conv2d_s2_wgrad(x, dz, self._g[f"conv{i}.kernel"], False, self.dtype, x16=x16, dy16=dz16)
            # the gradient handed to the next (lower) level's norm backward stays bf16 in the bf16 path; the image
            # gradient of level 1 is fp32 (consumed by the loss / tanh backward)
            if not want_dx:
                g_h = None
            elif m16 and i > 1 and drop32 and i - 1 >= lowest:
                # bf16 path: the conv that writes the gradient of level i-1 also adds up the first-pass sums of that level's
                # norm backward (its z and statistics are at hand) where the kernel covers the shape
                zl, stl = ctx["enc"][i - 2][1], ctx["enc"][i - 2][2]
                if rows is not None:
                    zl, stl = zl[rows], stl[rows]
                g_h, nfp = ops.conv2d_s2_dgrad(None, packs[i - 1], cb, self.dtype, dy16=dz16, out_bf16=True, fuse=(zl, stl, a))
            else:
                g_h = ops.conv2d_s2_dgrad(dz, packs[i - 1], cb, self.dtype, dy16=dz16, out_bf16=(self.dtype == DT_BF16 and i > 1))
        return g_h if need_input_grad else None


class Decoder(_ConvStack):
    """model.py:30-51.  conv_i: Conv2DTranspose(conv_filter[i], 5, (2,2), 'same') -> InstanceNorm -> leaky,
    with `x += add[i-1]` in front of every level whose skip is not None.
    Returns (x, x16): the fp32 output and/or its bf16 mirror (bf16 path: possibly only the mirror)."""

    def __init__(self, args):
        cf = args.conv_filter
        super().__init__(args, [(cf[i], cf[i - 1]) for i in range(1, 5)])  # (cb=out, cs=in)

    @staticmethod
    def _bias_dim(cb, cs):
        return cb

    def __call__(self, inputs, ctx: Optional[dict] = None, raw_last: bool = False):
        """raw_last (bf16 path, when the final layer can normalise while it stages — ops.convT_s1_tanh_fwd_z16_supported): the
        last level's InstanceNorm + LeakyReLU pass is NOT run; returns (None, None) and leaves that level's raw (z16, stats) in
        ctx["dec"][3] for the consumer.  Only for callers whose tapes never need the normalised level-4 map: it is the operand
        of the final layer's WEIGHT gradient, so the Generator keeps it; the Adjuster (trains its dense + norm only) does not."""
        x, add = inputs
        a = self.args.leaky_alpha
        packs = self.packs()
        saved = []
        m16 = self.dtype == DT_BF16
        # level 1's input: `x` is the dense + norm output with the first skip (add[0]) ALREADY added by the producer's apply
        # launch (_DenseNorm.__call__(skip=...), model.py:44-46 `x += add[0]`), given as (fp32, bf16 mirror | None)
        if isinstance(x, tuple):
            x, x16 = x
        else:
            if add[0] is not None:
                raise ValueError("Decoder: pass the level-1 input as the pair _DenseNorm.__call__(..., skip=add[0]) returns")
            x16 = None
        if m16 and x16 is None:
            raise ValueError("Decoder: the bf16 path needs the bf16 mirror of its input (_DenseNorm.__call__(want16=True))")
        for i, (cb, cs) in enumerate(self.chans, 1):
            gm, bt = self._w[f"norm{i}.gamma"], self._w[f"norm{i}.beta"]
            z, st = ops.convT_s2_fwd_stats(x, packs[i - 1], self._w[f"conv{i}.bias"], cb, self.dtype, gm, bt, x16=x16,
                                           z16=m16, alpha=a, defer_stats=m16 and _DEFER)
            if st is None:
                st = ops.instnorm_stats(z, gm, bt, 0, a)
            skip = add[i] if i < 4 else None
            # levels 1-3 feed only the next transposed conv and its weight gradient, level 4 the final 3-channel
            # layer: where those kernels read the bf16 mirror, the fp32 copy is not written at all
            if i < 4:
                drop32 = m16 and ops.conv_halo_supported(1, self.dtype, z.shape[0], z.shape[1], z.shape[2], cb,
                                                         self.chans[i][0])
                want16 = m16
            else:
                drop32 = want16 = m16 and ops.n3_m16_supported(z.shape[1], z.shape[2], self.args.image_channel, cb,
                                                               self.dtype)
            if i == 4 and raw_last and m16 and z.dtype == torch.bfloat16:
                st = ops.stats_tensor(st)   # no apply pass here: the stand-alone finalize
                saved.append((x, z, st, x16))
                x, x16 = None, None
                break
            h16 = torch.empty(z.shape, dtype=torch.bfloat16, device=z.device) if want16 else None
            if isinstance(skip, tuple):  # skip given as (first rows, remaining rows) of the batch: one apply per part
                b1 = skip[0].shape[0]
                h = None if drop32 else torch.empty(z.shape, dtype=torch.float32, device=z.device)
                for lo, hi, sk in ((0, b1, skip[0]), (b1, z.shape[0], skip[1])):
                    ops.instnorm_apply(z[lo:hi], st[lo:hi], sk, 0, 1, a, out=None if drop32 else h[lo:hi],
                                       out16=h16[lo:hi] if h16 is not None else None, want_f32=not drop32)
            else:
                h = ops.instnorm_apply(z, st, skip, 0, 1, a, out16=h16, want_f32=not drop32)
            st = ops.stats_tensor(st)   # deferred moments: finished by the apply launch(es) above
            saved.append((x, z, st, x16))
            x, x16 = h, h16
        if ctx is not None:
            ctx["dec"] = saved
        return x, x16

    def backward(self, ctx, g_h, need_wgrad: bool, wgrad_levels=None, need_input_grad: bool = True, partials=None):
        """partials: first-pass sums of level 4's norm backward if the conv that wrote g_h produced them (ops.NormPartials).
        g_h: fp32 or (bf16 path) bf16.  Returns the gradient w.r.t. the decoder input x (skip tensors receive the same gradient as the
        level input they were added to; no tape of the step asks for it), or None if need_input_grad is False.
        wgrad_levels / need_input_grad: as Encoder.backward — the chain stops at the lowest level anything is asked of."""
        a = self.args.leaky_alpha
        packs = self.packs()
        levels = set(range(1, 5)) if (need_wgrad and wgrad_levels is None) else (set(wgrad_levels or ()) if need_wgrad else set())
        lowest = 1 if need_input_grad else (min(levels) if levels else 5)
        any_wgrad = need_wgrad
        nfp = partials
        for i in range(4, lowest - 1, -1):
            need_wgrad = any_wgrad and i in levels
            cb, cs = self.chans[i - 1]
            x, z, st, x16 = ctx["dec"][i - 1]
            dgm = self._g[f"norm{i}.gamma"] if need_wgrad else None
            dbt = self._g[f"norm{i}.beta"] if need_wgrad else None
            want_dx = i > lowest or (i == 1 and need_input_grad)
            # A level that is asked for NO weight gradient (the Adjuster's tapes, eager_trainer.py:158-163; partition steps) has one
            # reader of its dz: the data-gradient conv below.  Where that kernel can take the norm backward through its operand
            # staging (ops.convT_s2_dgrad_bn: dz = a (g' - m1 - c m2') formed from (z, g) per halo piece), the apply pass and the
            # dz tensor are left out; only the per-sample coefficients are finished from the producer-fused sums.
            if (_BN and not need_wgrad and want_dx and i > 1 and i - 1 >= lowest and nfp is not None and self.dtype == DT_BF16
                    and z.dtype == torch.bfloat16 and g_h.dtype == torch.bfloat16
                    and ops.convT_s2_dgrad_bn_supported(z.shape[0], z.shape[1] // 2, z.shape[2] // 2, cb, cs, self.dtype)):
                coef = ops.instnorm_bwd_coef(z, st, nfp)
                zl, stl = ctx["dec"][i - 2][1], ctx["dec"][i - 2][2]
                g_h, nfp = ops.convT_s2_dgrad_bn(z, g_h, coef, a, packs[i - 1], cs, fuse=(zl, stl, a))
                continue
            dz16 = torch.empty(z.shape, dtype=torch.bfloat16, device=z.device) if self.dtype == DT_BF16 else None
            drop32 = (dz16 is not None and (not need_wgrad or x16 is not None) and
                      ops.conv_halo_supported(0, self.dtype, z.shape[0], z.shape[1] // 2, z.shape[2] // 2, cb, cs))
            dz = ops.instnorm_bwd(z, st, g_h, dgm, dbt, 0, 1, a, out16=dz16, want_f32=not drop32,
                                  db=self._g[f"conv{i}.bias"] if need_wgrad else None, partials=nfp)
            nfp = None
            if need_wgrad:
                ops.convT_s2_wgrad(x, dz, self._g[f"conv{i}.kernel"], False, self.dtype, x16=x16, dy16=dz16)
            if not want_dx:
                g_h = None
            elif dz16 is not None and drop32 and i > 1 and i - 1 >= lowest:
                zl, stl = ctx["dec"][i - 2][1], ctx["dec"][i - 2][2]  # the level the gradient belongs to
                g_h, nfp = ops.convT_s2_dgrad(None, packs[i - 1], cs, self.dtype, dy16=dz16, out_bf16=True, fuse=(zl, stl, a))
            else:
                # bf16 path: every data gradient of the decoder leaves its conv as bf16 — also level 1's, which the dense + norm
                # backward reads as fp32: the persistent bf16-output kernels (conv_down3.hip) then cover that level too (its
                # fp32-output route was the last user of the round-1 halo kernel here, 150-200 us per call against ~100); the dense
                # layer's norm backward takes the bf16 gradient as it is (lg_instnorm_leaky_bwd: g_is_bf16)
                g_h = ops.convT_s2_dgrad(dz, packs[i - 1], cs, self.dtype, dy16=dz16, out_bf16=self.dtype == DT_BF16)
        return g_h if need_input_grad else None


class _FinalConv(_Module):
    """Conv2DTranspose(image_channel, 5, strides 1, 'same', activation tanh)  model.py:86-87."""

    def __init__(self, args):
        super().__init__(args)
        self.cb, self.cs = args.image_channel, args.conv_filter[4]
        self._add("kernel", (5, 5, self.cb, self.cs), "kernel")
        self._add("bias", (self.cb,), "zeros")
        self._pack = None
        self._pack_version = -1
        self.version = 0

    def pack(self):
        if self._pack_version != self.version:
            self._pack = ops.conv_pack(self._w["kernel"], self.cb, self.cs, self.dtype, out=self._pack)
            self._pack_version = self.version
        return self._pack

    def __call__(self, x, out=None, x16=None):
        return ops.convT_s1_tanh_fwd(x, self.pack(), self._w["bias"], self.cb, self.dtype, out=out, x16=x16)

    def raw_supported(self, H, W):
        """the layer can take the decoder's RAW last map + its statistics (normalisation applied while staging)"""
        return self.dtype == DT_BF16 and ops.convT_s1_tanh_fwd_z16_supported(H, W, self.cb, self.cs, self.dtype)

    def from_raw(self, z16, stats, out=None):
        """tanh(convT_s1(LeakyReLU(InstanceNorm(z)))) with the normalised map never written (lg_convT_s1_tanh_fwd_z16,
        bit-identical to apply + conv: tests/test_ops_gpu.py)"""
        return ops.convT_s1_tanh_fwd_z16(z16, stats, self.args.leaky_alpha, self.pack(), self._w["bias"], self.cb, self.dtype, out=out)

    def backward(self, x, dpre, need_wgrad: bool, x16=None, need_dx: bool = True, fuse=None):
        """Returns dL/dx: fp32, or bf16 in the bf16 path (the decoder's norm backward reads either); None if not need_dx.
        fuse = (z16, stats, alpha) of the decoder's last level (bf16 path): returns (dx, NormPartials | None) instead."""
        B, H, W, _ = dpre.shape
        g16 = self.dtype == DT_BF16
        dx = torch.empty(B, H, W, self.cs, dtype=torch.bfloat16 if g16 else torch.float32, device=dpre.device) if need_dx else None
        if not need_dx and not need_wgrad:
            return (None, None) if fuse is not None else None
        use_fuse = fuse is not None and g16 and need_dx and fuse[0].dtype == torch.bfloat16
        r = ops.convT_s1_tanh_bwd(x if need_wgrad else None, dpre, self.pack(), self.cs, self.dtype,
                                  dx=None if (g16 or not need_dx) else dx, dx16=dx if (g16 and need_dx) else None,
                                  dw=self._g["kernel"] if need_wgrad else None,
                                  db=self._g["bias"] if need_wgrad else None, x16=x16 if need_wgrad else None,
                                  fuse=fuse if use_fuse else None)
        if fuse is not None:
            return (dx, r[1] if use_fuse else None)
        return dx


class _DenseNorm(_Module):
    """Dense(init_dim^2 * conv_filter[0]) -> leaky -> InstanceNormalization   (model.py:83-84,98-102 and
    :120-121,128-131; the norm sees a 4-D tensor in G and a 2-D one in A — same per-sample moments)."""

    def __init__(self, args, in_dim):
        super().__init__(args)
        self.in_dim = in_dim
        self.out_dim = args.init_dim ** 2 * args.conv_filter[0]
        self._add("dense.kernel", (in_dim, self.out_dim), "kernel")
        self._add("dense.bias", (self.out_dim,), "zeros")
        self._add("norm.gamma", (1,), "ones")
        self._add("norm.beta", (1,), "zeros")

    def __call__(self, x, ctx: Optional[dict] = None, skip=None, want16: bool = False):
        """Returns (w, w16): the [B, init_dim, init_dim, conv_filter[0]] map the decoder starts from, as fp32 and (want16) its
        bf16 mirror.  skip (optional): the decoder's first skip tensor, added in the SAME apply launch (model.py:44-46 `x += add[0]`),
        as a tensor or a pair (first rows, remaining rows) of the batch."""
        a = self.args.leaky_alpha
        u = ops.dense_fwd(x, self._w["dense.kernel"], self._w["dense.bias"])
        st = ops.instnorm_stats(u, self._w["norm.gamma"], self._w["norm.beta"], 1, a)
        w16 = torch.empty(u.shape, dtype=torch.bfloat16, device=u.device) if want16 else None
        if isinstance(skip, tuple):
            b1 = skip[0].shape[0]
            w = torch.empty_like(u)
            for lo, hi, sk in ((0, b1, skip[0]), (b1, u.shape[0], skip[1])):
                ops.instnorm_apply(u[lo:hi], st[lo:hi], sk.reshape(hi - lo, -1), 1, 0, a, out=w[lo:hi],
                                   out16=w16[lo:hi] if want16 else None)
        else:
            w = ops.instnorm_apply(u, st, skip.reshape(u.shape) if skip is not None else None, 1, 0, a, out16=w16)
        if ctx is not None:
            ctx["dn"] = (x, u, st)
        shp = (-1, self.args.init_dim, self.args.init_dim, self.args.conv_filter[0])
        return w.view(shp), (w16.view(shp) if want16 else None)

    def backward(self, ctx, g):
        x, u, st = ctx["dn"]
        a = self.args.leaky_alpha
        du = ops.instnorm_bwd(u, st, g.reshape(u.shape), self._g["norm.gamma"], self._g["norm.beta"], 1, 0, a)
        ops.dense_wgrad(x, du, self._g["dense.kernel"], self._g["dense.bias"])


class Generator(_Module):
    """model.py:76-105."""

    def __init__(self, args, decoder: Decoder):
        super().__init__(args)
        self._dn = _DenseNorm(args, args.noise_dim + args.cond_dim)
        self.decoder = decoder
        self.conv = _FinalConv(args)

    # reference attribute names (model.py:83-86)
    @property
    def dense(self):
        return self._dn

    @property
    def norm(self):
        return self._dn

    def parts(self):
        return [("gen.", self._dn), ("dec.", self.decoder), ("gen.conv.", self.conv)]

    @property
    def weights(self):
        return self._dn.weights + self.decoder.weights + self.conv.weights

    def __call__(self, inputs, ctx: Optional[dict] = None, out=None):
        noise, cond = inputs
        x0 = ops.concat_cols(noise, cond)
        w4 = self._dn(x0, ctx, want16=self.dtype == DT_BF16)
        xdec, xdec16 = self.decoder([w4, [None] * 4], ctx)
        img = self.conv(xdec, out=out, x16=xdec16)
        if ctx is not None:
            ctx["xdec"], ctx["xdec16"], ctx["img"] = xdec, xdec16, img
        return img

    def backward(self, ctx, dpre, train_range=None):
        """dpre = dL/d(pre-tanh image).  Writes the weight gradients of Generator.weights[lo:hi] (train_range; default all
        22): 0-3 dense + norm, 4i..4i+3 decoder level i, 20-21 final conv — the partition groups of eager_trainer.py:49."""
        lo, hi = train_range if train_range is not None else (0, 22)
        final = lo <= 20 and hi >= 22
        levels = [i for i in range(1, 5) if lo <= 4 * i and 4 * i + 4 <= hi]
        dn = lo <= 0 and hi >= 4
        below_final = dn or bool(levels)
        z4, st4 = ctx["dec"][3][1], ctx["dec"][3][2]
        g, nfp = self.conv.backward(ctx["xdec"], dpre, need_wgrad=final, x16=ctx.get("xdec16"), need_dx=below_final,
                                    fuse=(z4, st4, self.args.leaky_alpha))
        if below_final:
            g = self.decoder.backward(ctx, g, need_wgrad=bool(levels), wgrad_levels=levels, need_input_grad=dn, partials=nfp)
            if dn:
                self._dn.backward(ctx, g)


class _DenseHead:
    """`Dense(n, activation=sigmoid)` attribute of the reference's Discriminator (model.py:62-63) as a view of the
    weights the fused heads kernels use: .kernel, .bias, .weights (keras order) and a call y = sigmoid(x W + b)."""

    def __init__(self, owner: "_Module", name: str):
        self._o, self._n = owner, name

    @property
    def kernel(self):
        return self._o._w[self._n + ".kernel"]

    @property
    def bias(self):
        return self._o._w[self._n + ".bias"]

    @property
    def weights(self):
        return [self.kernel, self.bias]

    def __call__(self, x):
        o = self._o  # both heads come out of one fused kernel; this view returns its own columns
        p = ops.heads_fwd(x.contiguous(), o._w["dense_pr.kernel"], o._w["dense_pr.bias"], o._w["dense_cond.kernel"],
                          o._w["dense_cond.bias"])
        return p[:, :1] if self._n == "dense_pr" else p[:, 1:]


class Discriminator(_Module):
    """model.py:54-73.  Returns (output_pr [B,1], output_cond [B,cond_dim]) as views of one [B,1+c] buffer."""

    def __init__(self, args, encoder: Encoder):
        super().__init__(args)
        self.encoder = encoder
        k = args.init_dim ** 2 * args.conv_filter[0]
        self._add("dense_pr.kernel", (k, 1), "kernel")
        self._add("dense_pr.bias", (1,), "zeros")
        self._add("dense_cond.kernel", (k, args.cond_dim), "kernel")
        self._add("dense_cond.bias", (args.cond_dim,), "zeros")
        self.dense_pr = _DenseHead(self, "dense_pr")      # model.py:62
        self.dense_cond = _DenseHead(self, "dense_cond")  # model.py:63

    def parts(self):
        return [("enc.", self.encoder), ("disc.", self)]

    @property
    def weights(self):
        return self.encoder.weights + [self._w[n] for n in self._names]

    def forward_packed(self, image, ctx: Optional[dict] = None, keep_maps: bool = True, top_only: bool = False):
        """top_only: no weight gradient will be asked of this pass and only the heads read the encoder (see Encoder.__call__)."""
        outs = self.encoder(image, ctx, keep_maps=keep_maps, top_only=top_only)
        x = outs[3].view(image.shape[0], -1)
        p = ops.heads_fwd(x, self._w["dense_pr.kernel"], self._w["dense_pr.bias"], self._w["dense_cond.kernel"],
                          self._w["dense_cond.bias"])
        if ctx is not None:
            ctx["heads_x"] = x
        return p

    def __call__(self, inputs, ctx: Optional[dict] = None):
        p = self.forward_packed(inputs, ctx)
        return p[:, :1], p[:, 1:]

    def backward(self, ctx, dz, need_wgrad: bool, need_input_grad: bool, rows: Optional[slice] = None, train_range=None):
        """dz [B,1+c] = dL/d(logits).  need_wgrad: the disc tape (gradients of Discriminator.weights[lo:hi], train_range,
        default all 20: 4(i-1)..4i-1 encoder level i, 16-19 the heads); otherwise only the data path down to the image
        (gen / adj tapes)."""
        lo, hi = train_range if train_range is not None else (0, 20)
        heads = need_wgrad and lo <= 16 and hi >= 20
        levels = [i for i in range(1, 5) if lo <= 4 * (i - 1) and 4 * i <= hi] if need_wgrad else []
        x = ctx["heads_x"] if rows is None else ctx["heads_x"][rows]
        if heads:
            ops.heads_wgrad(x, dz, self._g["dense_pr.kernel"], self._g["dense_pr.bias"], self._g["dense_cond.kernel"],
                            self._g["dense_cond.bias"])
        if not levels and not need_input_grad:
            return None  # a heads-only partition step: nothing flows into the encoder
        dx = ops.heads_dgrad(dz, self._w["dense_pr.kernel"], self._w["dense_cond.kernel"])
        i, f = self.args.init_dim, self.args.conv_filter[0]
        return self.encoder.backward(ctx, dx.view(-1, i, i, f), bool(levels), need_input_grad, rows, wgrad_levels=levels)


class Adjuster(_Module):
    """model.py:108-136: encoder shared with D, decoder and final conv shared with G; own = dense, norm."""

    def __init__(self, args, discriminator: Discriminator, generator: Generator):
        super().__init__(args)
        self.encoder = discriminator.encoder
        self._dn = _DenseNorm(args, args.cond_dim)
        self.decoder = generator.decoder
        self.conv = generator.conv

    @property
    def dense(self):
        return self._dn

    @property
    def norm(self):
        return self._dn

    def parts(self):
        return [("adj.", self._dn)]

    @property
    def weights(self):
        return self.encoder.weights + self._dn.weights + self.decoder.weights + self.conv.weights

    def __call__(self, inputs, ctx: Optional[dict] = None, enc_tails=None):
        image, cond = inputs
        if enc_tails is None:
            enc = self.encoder(image)  # no context: no tape of the step differentiates through it
        else:  # the trailing samples of `image` were already encoded by D in this step (same weights): reuse
            # `image` holds the leading samples only (or the whole batch: then its trailing rows are the ones already encoded)
            own = cond.shape[0] - enc_tails[0].shape[0]
            enc = self.encoder(image[:own], None, tails=enc_tails)
        # the decoder's first skip (the encoder's top map, model.py:131-135) is added by the dense-norm apply launch
        c4 = self._dn(cond, ctx, skip=enc[3], want16=self.dtype == DT_BF16)
        # the Adjuster's tape differentiates dense + norm only (eager_trainer.py:51,62,163): nothing ever reads the decoder's
        # normalised last map again, so the final layer takes the raw map and normalises while it stages (bf16 path)
        side = self.args.init_dim * 16
        raw = self.conv.raw_supported(side, side)
        dctx = ctx if ctx is not None else ({} if raw else None)
        x, x16 = self.decoder([c4, [None] + enc[::-1][1:]], dctx, raw_last=raw)
        if raw and x is None and x16 is None:
            img = self.conv.from_raw(dctx["dec"][3][1], dctx["dec"][3][2])
        else:
            img = self.conv(x, x16=x16)
        if ctx is not None:
            ctx["xdec"], ctx["xdec16"], ctx["img"] = x, x16, img
        return img

    def backward_own(self, ctx, dpre):
        """Gradient w.r.t. Adjuster.weights[16:20] only (eager_trainer.py:51,62,163)."""
        z4, st4 = ctx["dec"][3][1], ctx["dec"][3][2]
        g, nfp = self.conv.backward(None, dpre, need_wgrad=False, fuse=(z4, st4, self.args.leaky_alpha))
        g = self.decoder.backward(ctx, g, need_wgrad=False, partials=nfp)
        self._dn.backward(ctx, g)


# ----------------------------------------------------------------------------------------------
class ParamStore:
    """Re-homes every weight of (G, D, A-own) into ONE flat fp32 buffer, in the order
    [Generator.weights (22) | Discriminator.weights (20) | Adjuster.weights[16:20] (4)], each weight
    16-byte aligned, with same-layout gradient and Adam-slot buffers.  The three optimizers, the
    partition groups (eager_trainer.py:48-52) and the data-parallel all-reduce buckets then are
    contiguous ranges of it."""

    def __init__(self, generator: Generator, discriminator: Discriminator, adjuster: Adjuster):
        self.models = {"G": generator, "D": discriminator, "A": adjuster}
        entries = []  # (model, name, module, local name, numel, shape)
        for m, mod in self.models.items():
            for prefix, part in mod.parts():
                for name, t in part.own_items():
                    entries.append((m, prefix + name, part, name, t.numel(), tuple(t.shape)))
        off = 0
        self.index = []
        for m, full, part, name, n, shp in entries:
            self.index.append((m, full, part, name, off, n, shp))
            off += (n + 3) // 4 * 4
        dev = generator.device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.ranges = {}  # model -> [(start, end_padded)] per weight index
        for m, full, part, name, o, n, shp in self.index:
            view = self.flat[o:o + n].view(shp)
            view.copy_(part._w[name])
            part._w[name] = view
            part._g[name] = self.grad[o:o + n].view(shp)
            self.ranges.setdefault(m, []).append((o, o + (n + 3) // 4 * 4))
        self._stacks = [discriminator.encoder, generator.decoder, generator.conv]

    def model_range(self, m, idx_lo=None, idx_hi=None):
        r = self.ranges[m]
        lo = 0 if idx_lo is None else idx_lo
        hi = len(r) if idx_hi is None else idx_hi
        return r[lo][0], r[hi - 1][1]

    def names(self, m):
        return [full for mm, full, *_ in self.index if mm == m]

    def bump(self):
        """Weights changed: conv packs must be rebuilt before the next forward."""
        for s in self._stacks:
            s.version += 1
