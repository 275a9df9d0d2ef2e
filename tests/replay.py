"""Call-by-call replay of a training step against the oracle (test infrastructure).

`OpRecorder` wraps the functions of `littlegan_amd.ops` while a step runs and keeps, for every kernel call, CPU copies
of its tensor arguments before and after the call.  `check_call` then recomputes that ONE call with the fp64 numpy
oracle from the recorded inputs and compares it with what the kernel wrote.

Why per call: in the bf16 configuration every layer rounds its operands and its stored output to bfloat16.  Two
implementations that differ by 1e-6 before such a rounding differ by one bf16 ulp on a fraction of the elements after
it, the next layer turns that into more flipped roundings, and after three layers the two are decorrelated at the level
of the bf16 rounding noise itself (measured: whole-step gradients of the kernels vs the bf16-emulating oracle agree to
1e-2 .. 3e-2 rms, against 1e-4 .. 1e-5 for any single call).  A whole-step comparison therefore cannot be tighter than
~1e-2 in bf16, whatever the kernels do; fed with the kernels' OWN inputs, each call can be held to 1e-3 and below, and
the chain of calls covers every tensor of the step at the step's real shapes.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import np_oracle as O

EPS = 1e-3  # InstanceNormalization epsilon (instance.py:47-58)

RECORDED = ["conv_pack", "conv2d_s2_fwd_stats", "conv2d_s2_fwd_stats_zn", "convT_s2_fwd_stats", "conv2d_s2_dgrad", "convT_s2_dgrad", "conv2d_s2_wgrad",
            "convT_s2_wgrad", "convT_s1_tanh_fwd", "convT_s1_tanh_fwd_z16", "convT_s1_tanh_bwd", "instnorm_stats", "instnorm_apply", "instnorm_bwd",
            "dense_fwd", "dense_wgrad", "heads_fwd", "heads_dgrad", "heads_wgrad", "instnorm_bwd_coef", "convT_s2_dgrad_bn"]


class _LazyStats:
    """deferred InstanceNorm moments (ops.Moments / a row range): their statistics tensor exists once the consuming apply has
    run, so it is read when the call is CHECKED, after the recorded step"""

    def __init__(self, m, lo, hi):
        self.m, self.lo, self.hi = m, lo, hi

    def resolve(self):
        return self.m.stats[self.lo:self.hi].detach().to("cpu", copy=True)


def _resolve(v):
    if isinstance(v, _LazyStats):
        return v.resolve()
    if isinstance(v, (list, tuple)):
        return type(v)(_resolve(u) for u in v)
    if isinstance(v, dict):
        return {k: _resolve(u) for k, u in v.items()}
    return v


def _snap(v):
    from littlegan_amd import ops
    if isinstance(v, ops.Moments):
        return _LazyStats(v, 0, v.B)
    if isinstance(v, ops._MomentRows):
        return _LazyStats(v.m, v.lo, v.hi)
    if isinstance(v, ops.NormPartials):   # the shared workspace is overwritten by the next fused launch: copy the records now
        n = v.shape[0] * v.nparts * 2
        return ("norm_partials", v.buf[:n * 8].detach().to("cpu", copy=True).view(torch.float64).reshape(v.shape[0], v.nparts, 2), v.alpha, v.shape)
    if torch.is_tensor(v):
        return v.detach().to("cpu", copy=True)
    if isinstance(v, (list, tuple)):
        return type(v)(_snap(u) for u in v)
    return v


class OpRecorder:
    def __init__(self, names=RECORDED):
        self.names = names
        self.calls = []      # dict(name, args, kwargs, post_args, post_kwargs, ret)
        self.packs = {}      # data_ptr of a packed operand image -> (w fp64 numpy, cb, cs, dtype)

    def __enter__(self):
        from littlegan_amd import ops
        self._ops = ops
        self._orig = {n: getattr(ops, n) for n in self.names}
        for n in self.names:
            setattr(ops, n, self._wrap(n, self._orig[n]))
        return self

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(self._ops, n, f)

    def _wrap(self, name, fn):
        def f(*args, **kwargs):
            pre_a, pre_k = _snap(args), {k: _snap(v) for k, v in kwargs.items()}
            ptrs = {"args": [a.data_ptr() if torch.is_tensor(a) else None for a in args]}
            ret = fn(*args, **kwargs)
            rec = dict(name=name, args=pre_a, kwargs=pre_k, post_args=_snap(args),
                       post_kwargs={k: _snap(v) for k, v in kwargs.items()}, ret=_snap(ret), ptrs=ptrs)
            if name == "conv_pack":
                self.packs[ret.data_ptr()] = (pre_a[0].double().numpy(), args[1], args[2], args[3])
            self.calls.append(rec)
            return ret
        return f


# ---------------------------------------------------------------------------------------------------------------------
def _np(t):
    return None if t is None else t.double().numpy()


def _rms_rel(got, exp):
    got, exp = np.asarray(got, np.float64).ravel(), np.asarray(exp, np.float64).ravel()
    return float(np.sqrt(((got - exp) ** 2).mean()) / (np.sqrt((exp ** 2).mean()) + 1e-30))


def _max_rel(got, exp):
    got, exp = np.asarray(got, np.float64).ravel(), np.asarray(exp, np.float64).ravel()
    return float(np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30))


class Tol:
    """f32: values computed and stored in fp32.  st16: values stored as bf16 (compared with the ROUNDED oracle value, so
    only rounding flips remain).  acc16: fp32 results of contractions over bf16 operands (same rounded operands on both
    sides -> only the accumulation order differs)."""
    f32 = dict(rms=1e-5, mx=1e-4)
    acc16 = dict(rms=2e-5, mx=2e-4)
    st16 = dict(rms=6e-4, mx=1.6e-2)   # a flipped rounding is one bf16 ulp = 2^-8 .. 2^-7 of the value
    red = dict(rms=2e-5, mx=2e-5)      # scalar / column reductions (fp64 accumulation in the kernels)


def _cmp(tag, got, exp, tol, floor=0.0):
    """rms- and max-relative comparison; `floor` (absolute) guards tensors that are pure cancellation noise"""
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    assert got.shape == exp.shape, (tag, got.shape, exp.shape)
    scale = np.abs(exp).max()
    if scale <= floor:
        assert np.abs(got - exp).max() <= max(floor, 1e-30) * 4, (tag, "floor", np.abs(got - exp).max(), floor)
        return 0.0
    r, m = _rms_rel(got, exp), _max_rel(got, exp)
    assert r <= tol["rms"] and m <= tol["mx"], (tag, f"rms {r:.3e} (<= {tol['rms']:.0e}) max {m:.3e} (<= {tol['mx']:.0e})")
    return r


def _operand(x, x16, bf16):
    """what the MFMA reads: the bf16 mirror if given, else the fp32 tensor (rounded in-kernel in the bf16 configuration)"""
    if x16 is not None:
        return _np(x16)
    v = _np(x)
    return O.bf16_round(v) if bf16 else v


def _w(rec, packs, idx):
    w, cb, cs, dtype = packs[rec["ptrs"]["args"][idx]]
    return (O.bf16_round(w) if dtype == 1 else w), dtype


def _stats_ref(z):
    B = z.shape[0]
    zf = z.reshape(B, -1)
    mu = zf.mean(1)
    sigma = np.sqrt(((zf - mu[:, None]) ** 2).mean(1))
    return mu, sigma


def _check_stats(tag, st, mu, sigma, gamma, beta, tol=2e-6):
    st = _np(st)
    scale = np.maximum(np.abs(mu), sigma) + 1e-30
    assert np.abs((st[:, 0] + st[:, 4]) - mu).max() <= tol * scale.max(), (tag, "mean")
    assert np.abs(st[:, 1] - sigma).max() <= tol * sigma.max(), (tag, "sigma")
    a = float(gamma) / (sigma + EPS)
    assert np.abs(st[:, 2] - a).max() <= tol * np.abs(a).max(), (tag, "scale")
    assert np.abs(st[:, 3] - float(beta)).max() <= 1e-7 * (abs(float(beta)) + 1e-30) + 1e-12, (tag, "beta")


def _kernel_affine_f32(x, st, pre_leaky, alpha):
    """c and y = a*c + b exactly as norm.hip forms them in fp32 (no contraction): used for the LeakyReLU MASK only, so
    that a pre-activation within rounding of zero takes the same branch on both sides"""
    B = x.shape[0]
    xf = x.reshape(B, -1).astype(np.float32)
    if pre_leaky:
        xf = np.where(xf > 0, xf, np.float32(alpha) * xf).astype(np.float32)
    s = st.astype(np.float32)
    c = ((xf - s[:, 0:1]) - s[:, 4:5]).astype(np.float32)
    y = ((s[:, 2:3] * c).astype(np.float32) + s[:, 3:4]).astype(np.float32)
    return c, y


def check_call(rec, packs, stats=None):
    """Recomputes one recorded call with the oracle; raises AssertionError on a mismatch.  Returns a short tag."""
    n, a, k, pa, pk, ret = (rec["name"], _resolve(rec["args"]), _resolve(rec["kwargs"]), _resolve(rec["post_args"]),
                            _resolve(rec["post_kwargs"]), _resolve(rec["ret"]))
    if n == "conv_pack":
        return "pack"
    if n in ("conv2d_s2_fwd_stats", "convT_s2_fwd_stats"):
        x, bias, ch, dtype, gm, bt = a[0], a[2], a[3], a[4], a[5], a[6]
        w, _ = _w(rec, packs, 1)
        bf = dtype == 1
        xq = _operand(x, k.get("x16"), bf)
        z_ref = (O.conv2d(xq, w, _np(bias), 2) if n[4] == "2" else O.conv2d_transpose(xq, w, _np(bias), 2))
        z, st = ret
        tag = f"{n}{tuple(z.shape)}"
        if z.dtype == torch.bfloat16:
            _cmp(tag + " z16", _np(z), O.bf16_round(z_ref), Tol.st16)
        else:
            _cmp(tag + " z", _np(z), z_ref, Tol.acc16 if bf else Tol.f32)
        if st is not None:
            mu, sigma = _stats_ref(z_ref)
            _check_stats(tag, st, mu, sigma, _np(gm)[0], _np(bt)[0])
        return tag
    if n == "conv2d_s2_fwd_stats_zn":
        # the stride-2 conv fed with the RAW map of the level below: x = bf16(LeakyReLU(a (z - mu) + beta)) is formed while the halo is
        # staged and never written; reference = the same rounding, then the conv on the rounded operands
        zin, st_in, alpha, bias, gm, bt = a[0], _np(a[1]), a[2], a[4], a[7], a[8]
        w, _ = _w(rec, packs, 3)
        v = _np(zin)
        B = v.shape[0]
        mu = (st_in[:, 0] + st_in[:, 4]).reshape((B, 1, 1, 1))
        h = O.bf16_round(O.leaky(st_in[:, 2].reshape(mu.shape) * (v - mu) + st_in[:, 3].reshape(mu.shape), alpha))
        z_ref = O.conv2d(h, w, _np(bias), 2)
        z, st = ret
        tag = f"{n}{tuple(z.shape)}"
        _cmp(tag + " z16", _np(z), O.bf16_round(z_ref), dict(rms=2e-3, mx=3e-2))   # (operand bits flip where a (z - mu) + beta sits on a bf16 tie)
        mu_r, sigma_r = _stats_ref(z_ref)
        _check_stats(tag, st, mu_r, sigma_r, _np(gm)[0], _np(bt)[0], tol=1e-4)
        return tag
    if n == "instnorm_stats":
        x, gm, bt, pre, alpha = a[0], a[1], a[2], a[3], a[4]
        v = _np(x)
        if pre:
            v = O.leaky(v, alpha)
        mu, sigma = _stats_ref(v)
        _check_stats(n, ret, mu, sigma, _np(gm)[0], _np(bt)[0])
        if pk.get("x16_out") is not None:
            _cmp(n + " x16_out", _np(pk["x16_out"]), O.bf16_round(_np(x)), dict(rms=1e-12, mx=1e-12))
        return n
    if n == "instnorm_apply":
        x, st, skip, pre, post, alpha = a[0], _np(a[1]), a[2], a[3], a[4], a[5]
        v = _np(x)
        B = v.shape[0]
        if pre:
            v = O.leaky(v, alpha)
        mu = (st[:, 0] + st[:, 4]).reshape((B,) + (1,) * (v.ndim - 1))
        y = st[:, 2].reshape(mu.shape) * (v - mu) + st[:, 3].reshape(mu.shape)
        if post:
            y = O.leaky(y, alpha)
        if skip is not None:
            y = y + _np(skip).reshape(y.shape)
        tag = f"{n}{tuple(x.shape)}"
        out = pk.get("out") if pk.get("out") is not None else (ret if torch.is_tensor(ret) else None)
        if out is not None:
            _cmp(tag + " out", _np(out), y, dict(rms=3e-6, mx=1e-4))
        if pk.get("out16") is not None:
            _cmp(tag + " out16", _np(pk["out16"]), O.bf16_round(y), Tol.st16)
        return tag
    if n == "instnorm_bwd":
        x, st, g, pre, post, alpha = a[0], _np(a[1]), a[2], a[5], a[6], a[7]
        assert not k.get("accumulate", False)
        xv, gv = _np(x), _np(g)
        B = xv.shape[0]
        xf = xv.reshape(B, -1)
        gf = gv.reshape(B, -1)
        c32, y32 = _kernel_affine_f32(xf, st, pre, alpha)
        xx = O.leaky(xf, alpha) if pre else xf
        mu = (st[:, 0] + st[:, 4])[:, None]
        sigma, aa = st[:, 1][:, None], st[:, 2][:, None]
        c = xx - mu
        gp = np.where(y32 > 0, gf, alpha * gf) if post else gf
        s = sigma + EPS
        m1 = gp.mean(1, keepdims=True)
        m2 = (gp * c).mean(1, keepdims=True) / (s * sigma)
        d = aa * (gp - m1 - c * m2)
        if pre:
            d = np.where(xf > 0, d, alpha * d)
        tag = f"{n}{tuple(x.shape)}"
        out = pk.get("out") if pk.get("out") is not None else (ret if torch.is_tensor(ret) else None)
        if out is not None:
            _cmp(tag + " dx", _np(out).reshape(B, -1), d, dict(rms=2e-5, mx=2e-4))
        if pk.get("out16") is not None:
            _cmp(tag + " dx16", _np(pk["out16"]).reshape(B, -1), O.bf16_round(d), Tol.st16)
        gabs = np.abs(gp).sum()
        if pa[3] is not None:  # dgamma, dbeta (scalars: sums with cancellation -> error relative to the sum of magnitudes)
            dg, dbt = float((gp * c / s).sum()), float(gp.sum())
            assert abs(float(pa[3][0]) - dg) <= 2e-6 * float((np.abs(gp * c) / s).sum()) + 1e-12, (tag, "dgamma", float(pa[3][0]), dg)
            assert abs(float(pa[4][0]) - dbt) <= 2e-6 * gabs + 1e-12, (tag, "dbeta", float(pa[4][0]), dbt)
        if pk.get("db") is not None:
            C = x.shape[-1]
            db = d.reshape(-1, C).sum(0)
            lim = 3e-6 * np.abs(d.reshape(-1, C)).sum(0).max()
            assert np.abs(_np(pk["db"]) - db).max() <= lim, (tag, "db", np.abs(_np(pk["db"]) - db).max(), lim)
        return tag
    if n == "instnorm_bwd_coef":
        # the norm backward without its apply pass: per-sample coefficient records from the producer-fused sums
        z, st, parts = a[0], _np(a[1]), a[2]
        assert isinstance(parts, tuple) and parts[0] == "norm_partials"
        S = parts[1].numpy().sum(1)
        L = z[0].numel()
        sigma = st[:, 1]
        m1, m2 = S[:, 0] / L, S[:, 1] / L / ((sigma + EPS) * sigma)
        co = _np(ret)
        tag = f"{n}{tuple(z.shape)}"
        for got, exp in ((co[:, 0] + co[:, 1], st[:, 0] + st[:, 4]), (co[:, 2], st[:, 2]), (co[:, 3], st[:, 3]), (co[:, 4] + co[:, 6], m1), (co[:, 5] + co[:, 7], m2)):
            assert np.abs(got - exp).max() <= 1e-7 * np.abs(exp).max() + 1e-30, (tag, got, exp)
        return tag
    if n == "convT_s2_dgrad_bn":
        # data gradient fed with the level's raw (z, g): dz = bf16(a (g' - m1 - c m2')) is formed while the halo is staged and never
        # written; reference = the same rounding, then the conv on the rounded operand.  m1 is also checked against mean(g') here
        # (the sums come from the producer's epilogue).
        z, g, co, alpha, ch = _np(a[0]), _np(a[1]), _np(a[2]), a[3], a[5]
        w, _ = _w(rec, packs, 4)
        B = z.shape[0]
        sh = (B,) + (1,) * (z.ndim - 1)
        mu, aa, bb = (co[:, 0] + co[:, 1]).reshape(sh), co[:, 2].reshape(sh), co[:, 3].reshape(sh)
        m1, m2 = (co[:, 4] + co[:, 6]).reshape(sh), (co[:, 5] + co[:, 7]).reshape(sh)
        c32 = ((z.astype(np.float32) - co[:, 0].astype(np.float32).reshape(sh)) - co[:, 1].astype(np.float32).reshape(sh))
        y32 = (co[:, 2].astype(np.float32).reshape(sh) * c32).astype(np.float32) + co[:, 3].astype(np.float32).reshape(sh)
        gp = np.where(y32 > 0, g, alpha * g)
        tag = f"{n}{tuple(ret[0].shape)}"
        m1_ref = gp.reshape(B, -1).mean(1)
        assert np.abs(m1.reshape(B) - m1_ref).max() <= 2e-6 * np.abs(gp).reshape(B, -1).mean(1).max() + 1e-12, (tag, "m1", m1.reshape(B), m1_ref)
        dz = O.bf16_round(aa * (gp - m1 - (z - mu) * m2))
        ref = O.conv_fwd(dz, w, 2)
        _cmp(tag, _np(ret[0]), O.bf16_round(ref), dict(rms=2e-3, mx=3e-2))   # (operand bits flip where dz sits on a bf16 tie)
        return tag
    if n in ("conv2d_s2_dgrad", "convT_s2_dgrad"):
        if isinstance(ret, tuple):  # (gradient, fused first-pass sums of the next norm backward): the sums are checked
            ret = ret[0]            # through the dz / dgamma / dbeta of the instnorm_bwd call that consumes them
        dy, ch, dtype = a[0], a[2], a[3]
        w, _ = _w(rec, packs, 1)
        bf = dtype == 1
        dq = _operand(dy, k.get("dy16"), bf)
        if n[4] == "2":
            ref = O.conv_bwd_input(dq, w, 2, (2 * dq.shape[1], 2 * dq.shape[2]))
        else:
            ref = O.conv_fwd(dq, w, 2)
        tag = f"{n}{tuple(ret.shape)}"
        if ret.dtype == torch.bfloat16:
            _cmp(tag, _np(ret), O.bf16_round(ref), Tol.st16)
        else:
            _cmp(tag, _np(ret), ref, Tol.acc16 if bf else Tol.f32)
        return tag
    if n in ("conv2d_s2_wgrad", "convT_s2_wgrad"):
        x, dy, dtype = a[0], a[1], a[4]
        assert not a[3], "accumulate"
        bf = dtype == 1
        xq = _operand(x, k.get("x16"), bf)
        dq = _operand(dy, k.get("dy16"), bf)
        ref = O.conv_bwd_filter(xq, dq, 2, 5) if n[4] == "2" else O.conv_bwd_filter(dq, xq, 2, 5)
        tag = f"{n}{tuple(pa[2].shape)}"
        _cmp(tag, _np(pa[2]), ref, Tol.acc16 if bf else dict(rms=1e-5, mx=1e-4))
        return tag
    if n == "convT_s1_tanh_fwd":
        x, bias, dtype = a[0], a[2], a[4]
        w, _ = _w(rec, packs, 1)
        bf = dtype == 1
        xq = _operand(x, k.get("x16"), bf)
        ref = np.tanh(O.conv2d_transpose(xq, w, _np(bias), 1))
        out = pk.get("out") if pk.get("out") is not None else ret
        _cmp(n, _np(out), ref, dict(rms=2e-5, mx=2e-4) if bf else dict(rms=3e-6, mx=3e-5))
        return n
    if n == "convT_s1_tanh_fwd_z16":
        # the Adjuster's final layer fed with the RAW last decoder map: h = bf16(LeakyReLU(a (z - mu) + beta)) is formed while
        # staging and never written; reference = the same rounding, then the transposed conv on the rounded operands
        z16, st, alpha, bias = a[0], _np(a[1]), a[2], a[4]
        w, _ = _w(rec, packs, 3)
        v = _np(z16)
        B = v.shape[0]
        mu = (st[:, 0] + st[:, 4]).reshape((B, 1, 1, 1))
        h = O.bf16_round(O.leaky(st[:, 2].reshape(mu.shape) * (v - mu) + st[:, 3].reshape(mu.shape), alpha))
        ref = np.tanh(O.conv2d_transpose(h, w, _np(bias), 1))
        out = pk.get("out") if pk.get("out") is not None else ret
        _cmp(n, _np(out), ref, dict(rms=2e-4, mx=4e-3))   # (an a (z - mu) + beta at rounding distance from a bf16 tie flips one operand bit)
        return n
    if n == "convT_s1_tanh_bwd":
        x, dpre, cs, dtype = a[0], a[1], a[3], a[4]
        w, _ = _w(rec, packs, 2)
        bf = dtype == 1
        dq = O.bf16_round(_np(dpre)) if bf else _np(dpre)
        if pk.get("dx") is not None:
            _cmp(n + " dx", _np(pk["dx"]), O.conv_fwd(dq, w, 1), Tol.acc16 if bf else Tol.f32)
        if pk.get("dx16") is not None:
            _cmp(n + " dx16", _np(pk["dx16"]), O.bf16_round(O.conv_fwd(dq, w, 1)), Tol.st16)
        if pk.get("dw") is not None:
            xq = _operand(x, k.get("x16"), bf)
            _cmp(n + " dw", _np(pk["dw"]), O.conv_bwd_filter(dq, xq, 1, 5), Tol.acc16 if bf else dict(rms=1e-5, mx=1e-4))
        if pk.get("db") is not None:
            dp = _np(dpre)
            ref = dp.sum((0, 1, 2))
            assert np.abs(_np(pk["db"]) - ref).max() <= 3e-6 * np.abs(dp).sum((0, 1, 2)).max(), (n, "db")
        return n
    if n == "dense_fwd":
        out = pk.get("out") if pk.get("out") is not None else ret
        _cmp(n, _np(out), _np(a[0]) @ _np(a[1]) + _np(a[2]), Tol.f32)
        return n
    if n == "dense_wgrad":
        x, dy = _np(a[0]), _np(a[1])
        _cmp(n + " dw", _np(pa[2]), x.T @ dy, dict(rms=1e-5, mx=1e-4))
        assert np.abs(_np(pa[3]) - dy.sum(0)).max() <= 3e-6 * np.abs(dy).sum(0).max() + 1e-12, (n, "db")
        return n
    if n == "heads_fwd":
        x, wpr, bpr, wc, bc = (_np(v) for v in a[:5])
        ref = np.concatenate([O.sigmoid(x @ wpr + bpr), O.sigmoid(x @ wc + bc)], 1)
        _cmp(n, _np(ret), ref, dict(rms=3e-6, mx=3e-5))
        return n
    if n == "heads_dgrad":
        dz, wpr, wc = (_np(v) for v in a[:3])
        _cmp(n, _np(ret), dz[:, :1] @ wpr.T + dz[:, 1:] @ wc.T, dict(rms=1e-5, mx=1e-4))
        return n
    if n == "heads_wgrad":
        x, dz = _np(a[0]), _np(a[1])
        _cmp(n + " dwpr", _np(pa[2]), x.T @ dz[:, :1], dict(rms=1e-5, mx=1e-4))
        _cmp(n + " dwc", _np(pa[4]), x.T @ dz[:, 1:], dict(rms=1e-5, mx=1e-4))
        assert np.abs(_np(pa[3]) - dz[:, :1].sum(0)).max() <= 3e-6 * np.abs(dz).sum() + 1e-12
        assert np.abs(_np(pa[5]) - dz[:, 1:].sum(0)).max() <= 3e-6 * np.abs(dz).sum(0).max() + 1e-12
        return n
    raise KeyError(n)
