"""fp64 numpy ORACLE for the LittleGAN training-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``littlegan_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker.

PARITY UNPINNED: the reference's arithmetic lives in tensorflow-gpu==1.15.4
(``/root/reference/requirements.txt:2``), which is not installed in this
pipeline, and the reference ships no tests / golden vectors.
This file restates the published TF-1.15 semantics of the ops the reference
calls; it is cross-checked against an independent torch-autograd restatement
(``oracle/torch_oracle.py``) and the known-answer tests of SURVEY.md §8c.

Every function cites the reference lines (relative to /root/reference) it
follows.  All tensors are NHWC, float64.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F64 = np.float64


# --------------------------------------------------------------------------
# configuration (sample.config.json:1-54 — only the keys the hot path reads)
# --------------------------------------------------------------------------
@dataclass
class Cfg:
    batch_size: int = 32
    image_channel: int = 3
    noise_dim: int = 93
    init_dim: int = 8
    conv_filter: Tuple[int, ...] = (384, 256, 128, 64, 32)
    kernel_size: int = 5
    leaky_alpha: float = 0.3
    l1_lambda: float = 0.02
    lr: float = 5e-5
    beta_1: float = 0.5
    beta_2: float = 0.9
    use_clip: bool = True
    clip_range: float = 0.5
    use_partition: bool = True
    partition_interval: int = 4
    train_adj: bool = True
    cond_dim: int = 7
    # NOT a reference key: restate the bf16 MFMA configuration of the build under test (BASELINE.json configs[2]) —
    # round to bfloat16 (RNE) exactly where its kernels do, everything else stays fp64.  See bf16_round / _q below.
    emulate_bf16: bool = False

    @property
    def image_dim(self) -> int:  # model.py:83,101 + 4 stride-2 convT
        return self.init_dim * 16


# --------------------------------------------------------------------------
# utils.py:47-56
# --------------------------------------------------------------------------
def soft(x):
    """utils.py:47-48"""
    return 0.96 * x + 0.02


def data_rescale(x):
    """utils.py:51-52"""
    return np.asarray(x, F64) / 127.5 - 1.0


def inverse_rescale(y):
    """utils.py:55-56 (tf.round = round-half-to-even, like np.round)"""
    return np.round((np.asarray(y, F64) + 1.0) * 127.5)


# --------------------------------------------------------------------------
# bf16 emulation (test infrastructure for the bf16 MFMA path; the reference itself is fp32 throughout).
# The build's bf16 configuration rounds (round-to-nearest-even, fp32 -> bf16) at these points and nowhere else:
#   * every 5x5 conv / transposed-conv OPERAND: the layer input (activation mirror, 3-channel image, or the incoming
#     gradient dz / dpre of a data- or weight-gradient contraction) and the kernel;
#   * the raw conv output z as it is stored (the InstanceNorm MOMENTS are taken from the unrounded accumulators, the
#     normalisation and both norm-backward passes read the rounded z);
#   * the data gradients handed from one 5x5 layer to the next level's norm backward (not the image gradient, not the
#     gradient that leaves the decoder towards the dense layer).
# Accumulation (fp32 in the kernels) and everything outside the 5x5 layers (dense, heads, losses, Adam) is fp64 here.
# With this mode on, a bf16 kernel bug shows as an O(1e-2) mismatch instead of hiding under an O(1e-1) tolerance.
# --------------------------------------------------------------------------
def bf16_round(x):
    """fp64 -> fp32 (RNE) -> bf16 (RNE) -> fp64, elementwise."""
    a = np.ascontiguousarray(np.asarray(x, F64).astype(np.float32))
    u = a.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return r.view(np.float32).astype(F64)


def _q(cfg, x):
    return bf16_round(x) if getattr(cfg, "emulate_bf16", False) else x


TRACE = None  # diagnostics: set to a list to record (tag, array) for the backward intermediates


def _trace(tag, a):
    if TRACE is not None and a is not None:
        TRACE.append((tag, np.array(a)))


# --------------------------------------------------------------------------
# TF "SAME" convolution arithmetic (tf.compat.v1.layers.Conv2D / Conv2DTranspose,
# model.py:15, 39-40, 86-87).  TF-1.15: out = ceil(in/s);
# pad_total = max((out-1)*s + k - in, 0); pad_before = pad_total // 2.
# --------------------------------------------------------------------------
def same_pads(n_in: int, k: int, s: int) -> Tuple[int, int, int]:
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2, total - total // 2


def conv_fwd(x: np.ndarray, w: np.ndarray, s: int) -> np.ndarray:
    """NHWC conv, SAME, stride s, kernel HWIO [k,k,Ci,Co]  (model.py:15)."""
    B, H, W, Ci = x.shape
    k = w.shape[0]
    Ho, pt, pb = same_pads(H, k, s)
    Wo, pl, pr = same_pads(W, k, s)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((B, Ho, Wo, w.shape[3]), F64)
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, ky:ky + (Ho - 1) * s + 1:s, kx:kx + (Wo - 1) * s + 1:s, :]
            y += patch @ w[ky, kx]
    return y


def conv_bwd_input(dy: np.ndarray, w: np.ndarray, s: int, in_hw: Tuple[int, int]) -> np.ndarray:
    """Adjoint of conv_fwd w.r.t. x.  This IS tf.nn.conv2d_backprop_input, i.e. the
    forward of Conv2DTranspose when w is laid out HWOI (model.py:39-40)."""
    H, W = in_hw
    B, Ho, Wo, Co = dy.shape
    k = w.shape[0]
    Ho2, pt, pb = same_pads(H, k, s)
    Wo2, pl, pr = same_pads(W, k, s)
    assert (Ho2, Wo2) == (Ho, Wo), "conv_bwd_input: inconsistent geometry"
    dxp = np.zeros((B, H + pt + pb, W + pl + pr, w.shape[2]), F64)
    for ky in range(k):
        for kx in range(k):
            dxp[:, ky:ky + (Ho - 1) * s + 1:s, kx:kx + (Wo - 1) * s + 1:s, :] += dy @ w[ky, kx].T
    return dxp[:, pt:pt + H, pl:pl + W, :]


def conv_bwd_filter(x: np.ndarray, dy: np.ndarray, s: int, k: int) -> np.ndarray:
    """Adjoint of conv_fwd w.r.t. w -> [k,k,Ci,Co]."""
    B, H, W, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    _, pt, pb = same_pads(H, k, s)
    _, pl, pr = same_pads(W, k, s)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dw = np.zeros((k, k, Ci, Co), F64)
    dyf = dy.reshape(-1, Co)
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, ky:ky + (Ho - 1) * s + 1:s, kx:kx + (Wo - 1) * s + 1:s, :]
            dw[ky, kx] = patch.reshape(-1, Ci).T @ dyf
    return dw


def conv2d(x, w, b, s=2):
    """tf.compat.v1.layers.Conv2D(f, 5, 2, 'same')  (model.py:15)."""
    return conv_fwd(x, w, s) + b


def conv2d_bwd(x, w, dy, s=2):
    return (conv_bwd_input(dy, w, s, x.shape[1:3]),
            conv_bwd_filter(x, dy, s, w.shape[0]),
            dy.sum(axis=(0, 1, 2)))


def conv2d_transpose(x, w, b, s):
    """tf.compat.v1.layers.Conv2DTranspose(f, 5, (s,s), 'same'), kernel HWOI
    [k,k,Co,Ci]; output side = s*in  (model.py:39-40, 86-87)."""
    H, W = x.shape[1] * s, x.shape[2] * s
    return conv_bwd_input(x, w, s, (H, W)) + b


def conv2d_transpose_bwd(x, w, dy, s):
    dx = conv_fwd(dy, w, s)
    dw = conv_bwd_filter(dy, x, s, w.shape[0])  # [k,k,Co,Ci]
    return dx, dw, dy.sum(axis=(0, 1, 2))


# --------------------------------------------------------------------------
# elementwise
# --------------------------------------------------------------------------
def leaky(x, a):
    """tf.nn.leaky_relu (model.py:24,50,100,130)"""
    return np.where(x > 0, x, a * x)


def leaky_bwd(x, dy, a):
    """LeakyReluGrad: features > 0 ? g : alpha*g"""
    return np.where(x > 0, dy, a * dy)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------
# InstanceNormalization(axis=None, eps=1e-3)   instance.py:105-128
# per-sample moments over ALL non-batch axes; scalar gamma/beta (shape (1,));
# eps is added to the std (instance.py:115).
# --------------------------------------------------------------------------
IN_EPS = 1e-3


def instnorm(x, gamma, beta, eps=IN_EPS, xq=None):
    """xq (bf16 emulation only): the stored (rounded) copy of x that is normalised; the moments are those of x."""
    B = x.shape[0]
    xf = x.reshape(B, -1)
    mu = xf.mean(axis=1, keepdims=True)
    c = xf - mu
    sigma = np.sqrt((c * c).mean(axis=1, keepdims=True))  # K.std: population std
    if xq is not None:
        c = xq.reshape(B, -1) - mu
    s = sigma + eps
    y = gamma * (c / s) + beta
    return y.reshape(x.shape), (c, sigma, s)


# Scale of the two scalar gradients (sums with heavy cancellation): while a dict, instnorm_bwd appends per call
# (sum |t|, sqrt(sum t^2)) of the summands t of dgamma and of dbeta under the current tag — step_gradients maps them to weight
# indices (out["scalar_scale"]) so that a test can bound the error of such a sum against its summands, not its value.
_SCALE_LOG = None
_SCALE_TAG = None


def instnorm_bwd(cache, gamma, dy):
    c, sigma, s = cache
    B = dy.shape[0]
    dyf = dy.reshape(B, -1)
    if _SCALE_LOG is not None:
        tg = dyf * c / s
        _SCALE_LOG.setdefault(_SCALE_TAG, []).append(((float(np.abs(tg).sum()), float(np.sqrt((tg * tg).sum()))),
                                                      (float(np.abs(dyf).sum()), float(np.sqrt((dyf * dyf).sum())))))
    m1 = dyf.mean(axis=1, keepdims=True)
    m2 = (dyf * c).mean(axis=1, keepdims=True)
    dx = (gamma / s) * (dyf - m1 - c * m2 / (s * sigma))
    dgamma = float((dyf * c / s).sum())
    dbeta = float(dyf.sum())
    return dx.reshape(dy.shape), dgamma, dbeta


# --------------------------------------------------------------------------
# losses   eager_trainer.py:85-102 ; tf.keras.losses.binary_crossentropy in
# TF-1.15 = K.binary_crossentropy on probabilities:
#   p^ = clip(p, eps, 1-eps);  bce = -(t*log(p^+eps) + (1-t)*log(1-p^+eps)),  eps=1e-7
# then mean over the last axis; the trainer takes reduce_mean over the batch.
# --------------------------------------------------------------------------
# TF evaluates this in float32: epsilon = float32(1e-7) and the upper clip bound is float32(1) - epsilon
# = 0.99999988 (not 1 - 1e-7).  The oracle keeps those two float32 constants so that saturated
# probabilities (p == 0 or p == 1 in float32) are scored as TF would score them.
BCE_EPS = float(np.float32(1e-7))
BCE_HI = float(np.float32(1.0) - np.float32(1e-7))


def bce_mean(t, p):
    t = np.broadcast_to(np.asarray(t, F64), p.shape)
    pc = np.clip(p, BCE_EPS, BCE_HI)
    l = -(t * np.log(pc + BCE_EPS) + (1.0 - t) * np.log(1.0 - pc + BCE_EPS))
    return float(l.mean(axis=-1).mean())


def bce_mean_bwd(t, p):
    """d(bce_mean)/dp ; clip_by_value passes gradient only inside [eps, 1-eps]."""
    t = np.broadcast_to(np.asarray(t, F64), p.shape)
    pc = np.clip(p, BCE_EPS, BCE_HI)
    g = -(t / (pc + BCE_EPS) - (1.0 - t) / (1.0 - pc + BCE_EPS)) / p.size
    inside = (p >= BCE_EPS) & (p <= BCE_HI)
    return np.where(inside, g, 0.0)


def l1_mean(a, b):
    return float(np.abs(a - b).mean())


def l1_mean_bwd_b(a, b):
    """d mean|a-b| / db  (tf.abs grad = sign, sign(0)=0)"""
    return -np.sign(a - b) / a.size


# --------------------------------------------------------------------------
# weights.  Order = keras Model.weights order the trainer indexes into
# (eager_trainer.py:48-63): Generator 22, Discriminator 20, Adjuster 38 (own 16:20).
# --------------------------------------------------------------------------
def glorot_uniform(rng, shape):
    """TF default kernel_initializer of tf.compat.v1.layers.{Dense,Conv2D,Conv2DTranspose}."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(F64)


def weight_shapes(cfg: Cfg) -> Dict[str, List[Tuple[str, Tuple[int, ...]]]]:
    cf, k, c0 = cfg.conv_filter, cfg.kernel_size, cfg.init_dim ** 2 * cfg.conv_filter[0]
    enc, dec = [], []
    cin = cfg.image_channel
    for i in range(1, 5):  # model.py:13-16  conv_filter[4-i]
        co = cf[4 - i]
        enc += [(f"enc.conv{i}.kernel", (k, k, cin, co)), (f"enc.conv{i}.bias", (co,)),
                (f"enc.norm{i}.gamma", (1,)), (f"enc.norm{i}.beta", (1,))]
        cin = co
    cin = cf[0]
    for i in range(1, 5):  # model.py:37-41  conv_filter[i], kernel HWOI
        co = cf[i]
        dec += [(f"dec.conv{i}.kernel", (k, k, co, cin)), (f"dec.conv{i}.bias", (co,)),
                (f"dec.norm{i}.gamma", (1,)), (f"dec.norm{i}.beta", (1,))]
        cin = co
    final = [("gen.conv.kernel", (k, k, cfg.image_channel, cf[4])), ("gen.conv.bias", (cfg.image_channel,))]
    G = [("gen.dense.kernel", (cfg.noise_dim + cfg.cond_dim, c0)), ("gen.dense.bias", (c0,)),
         ("gen.norm.gamma", (1,)), ("gen.norm.beta", (1,))] + dec + final  # model.py:83-86
    D = enc + [("disc.dense_pr.kernel", (c0, 1)), ("disc.dense_pr.bias", (1,)),
               ("disc.dense_cond.kernel", (c0, cfg.cond_dim)), ("disc.dense_cond.bias", (cfg.cond_dim,))]
    A_own = [("adj.dense.kernel", (cfg.cond_dim, c0)), ("adj.dense.bias", (c0,)),
             ("adj.norm.gamma", (1,)), ("adj.norm.beta", (1,))]  # model.py:120-121
    return {"G": G, "D": D, "A": A_own}


def init_weights(cfg: Cfg, seed: int = 0) -> Dict[str, List[np.ndarray]]:
    rng = np.random.default_rng(seed)
    out = {}
    for m, lst in weight_shapes(cfg).items():
        ws = []
        for name, shp in lst:
            if name.endswith("kernel"):
                ws.append(glorot_uniform(rng, shp))
            elif name.endswith("gamma"):
                ws.append(np.ones(shp, F64))
            else:
                ws.append(np.zeros(shp, F64))
        out[m] = ws
    return out


def adjuster_weights(W) -> List[np.ndarray]:
    """Adjuster.weights (38): encoder(16), dense(2), norm(2), decoder(16), conv(2)
    (model.py:119-123 attribute order)."""
    return W["D"][0:16] + W["A"] + W["G"][4:20] + W["G"][20:22]


PART_GROUPS = {  # eager_trainer.py:48-52
    "G": [range(0, 4), range(4, 8), range(8, 22)],
    "D": [range(0, 12), range(12, 16), range(16, 20)],
    "A": [range(0, 4)],  # = Adjuster.weights[16:20]
}


def train_weight_indices(cfg: Cfg, model: str, batch_no: int) -> List[int]:
    """eager_trainer.py:104-113 ; indices into W[model]."""
    n_all = {"G": 22, "D": 20, "A": 4}[model]
    if cfg.use_partition and batch_no % (cfg.partition_interval + 1) == 0:
        groups = PART_GROUPS[model]
        return list(groups[(batch_no // (cfg.partition_interval + 1)) % len(groups)])
    return list(range(n_all))


# --------------------------------------------------------------------------
# model forward / backward
# --------------------------------------------------------------------------
def encoder_fwd(cfg, We, x):
    """Encoder.call model.py:18-27 (dropout at :25 is identity: training=False default)."""
    outs, caches = [], []
    emu = getattr(cfg, "emulate_bf16", False)
    for i in range(4):
        k, b, g, be = We[4 * i:4 * i + 4]
        x = _q(cfg, x)  # MFMA operand (bf16 emulation; identity otherwise)
        z = conv2d(x, _q(cfg, k), b, 2)
        y, nc = instnorm(z, g[0], be[0], xq=_q(cfg, z) if emu else None)
        h = leaky(y, cfg.leaky_alpha)
        if emu and i < 3:
            h = bf16_round(h)  # maps 1-3 exist only as the bf16 mirrors (next conv operand, Adjuster skip); map 4 is fp32
        caches.append((x, y, nc))
        outs.append(h)
        x = h
    return outs, caches


def encoder_bwd(cfg, We, caches, d_outs, need_wgrad=True, need_input_grad=False):
    """d_outs: list of 4 grads w.r.t. the 4 returned maps (None = zero)."""
    grads = [None] * 16
    g_h = None
    for i in reversed(range(4)):
        k, b, g, be = We[4 * i:4 * i + 4]
        x, y, nc = caches[i]
        if d_outs[i] is not None:
            g_h = d_outs[i] if g_h is None else g_h + d_outs[i]
        dy = leaky_bwd(y, g_h, cfg.leaky_alpha)
        dz, dg, dbe = instnorm_bwd(nc, g[0], dy)
        dzq = _q(cfg, dz)  # MFMA operand of the data- and weight-gradient contractions; the bias sums use dz itself
        want_dx = (i > 0) or need_input_grad
        dx = conv_bwd_input(dzq, _q(cfg, k), 2, x.shape[1:3]) if want_dx else None
        if dx is not None and i > 0:
            dx = _q(cfg, dx)  # inter-layer gradients are stored as bf16; the image gradient (level 1) is fp32
        _trace(f"enc{i + 1}.dz", dz)
        _trace(f"enc{i + 1}.dx", dx)
        if need_wgrad:
            grads[4 * i] = conv_bwd_filter(x, dzq, 2, k.shape[0])
            grads[4 * i + 1] = dz.sum(axis=(0, 1, 2))
            grads[4 * i + 2] = np.array([dg])
            grads[4 * i + 3] = np.array([dbe])
        g_h = dx
    return grads, g_h


def decoder_fwd(cfg, Wd, x, add):
    """Decoder.call model.py:43-51."""
    caches = []
    for i in range(4):
        k, b, g, be = Wd[4 * i:4 * i + 4]
        if add[i] is not None:
            x = x + add[i]
        x = _q(cfg, x)  # MFMA operand
        z = conv2d_transpose(x, _q(cfg, k), b, 2)
        y, nc = instnorm(z, g[0], be[0], xq=_q(cfg, z) if getattr(cfg, "emulate_bf16", False) else None)
        h = leaky(y, cfg.leaky_alpha)
        caches.append((x, y, nc))
        x = h
    return x, caches


def decoder_bwd(cfg, Wd, caches, g_h, need_wgrad=True):
    """Returns (grads[16], grad w.r.t. decoder input x0).  Grads w.r.t. the skip
    tensors equal the running input grad at each level and are not needed by
    any gradient set the trainer requests (eager_trainer.py:145,149,163)."""
    grads = [None] * 16
    for i in reversed(range(4)):
        k, b, g, be = Wd[4 * i:4 * i + 4]
        x, y, nc = caches[i]
        dy = leaky_bwd(y, g_h, cfg.leaky_alpha)
        dz, dg, dbe = instnorm_bwd(nc, g[0], dy)
        dx, dw, _ = conv2d_transpose_bwd(x, _q(cfg, k), _q(cfg, dz), 2)
        db = dz.sum(axis=(0, 1, 2))
        dx = _q(cfg, dx)  # (bf16 emulation) every data gradient of the decoder is stored as bf16, the one handed to the dense layer too
        _trace(f"dec{i + 1}.dz", dz)
        _trace(f"dec{i + 1}.dx", dx)
        if need_wgrad:
            grads[4 * i:4 * i + 4] = [dw, db, np.array([dg]), np.array([dbe])]
        g_h = dx
    return grads, g_h


def generator_fwd(cfg, Wg, noise, cond):
    """Generator.call model.py:89-105."""
    x0 = np.concatenate([noise, cond], axis=-1)
    u = x0 @ Wg[0] + Wg[1]
    v = leaky(u, cfg.leaky_alpha)
    v4 = v.reshape(-1, cfg.init_dim, cfg.init_dim, cfg.conv_filter[0])
    w, nc = instnorm(v4, Wg[2][0], Wg[3][0])
    xdec, dcaches = decoder_fwd(cfg, Wg[4:20], w, [None] * 4)
    xdec = _q(cfg, xdec)
    pre = conv2d_transpose(xdec, _q(cfg, Wg[20]), Wg[21], 1)
    img = np.tanh(pre)
    return img, (x0, u, nc, dcaches, xdec, img)


def generator_bwd(cfg, Wg, cache, d_img):
    x0, u, nc, dcaches, xdec, img = cache
    grads = [None] * 22
    dpre = d_img * (1.0 - img * img)
    dxdec, dwf, _ = conv2d_transpose_bwd(xdec, _q(cfg, Wg[20]), _q(cfg, dpre), 1)
    dxdec = _q(cfg, dxdec)
    _trace("final.dpre", dpre)
    _trace("final.dx", dxdec)
    grads[20], grads[21] = dwf, dpre.sum(axis=(0, 1, 2))
    dgrads, dw4 = decoder_bwd(cfg, Wg[4:20], dcaches, dxdec)
    grads[4:20] = dgrads
    dv4, dg, dbe = instnorm_bwd(nc, Wg[2][0], dw4)
    grads[2], grads[3] = np.array([dg]), np.array([dbe])
    du = leaky_bwd(u, dv4.reshape(u.shape), cfg.leaky_alpha)
    grads[0] = x0.T @ du
    grads[1] = du.sum(axis=0)
    return grads


def discriminator_fwd(cfg, Wd, image):
    """Discriminator.call model.py:65-73."""
    outs, ecaches = encoder_fwd(cfg, Wd[0:16], image)
    x = outs[3].reshape(image.shape[0], -1)
    pr = sigmoid(x @ Wd[16] + Wd[17])
    c = sigmoid(x @ Wd[18] + Wd[19])
    return (pr, c), (ecaches, x, pr, c, outs[3].shape)


def discriminator_bwd(cfg, Wd, cache, d_pr, d_c, need_wgrad=True, need_input_grad=False):
    ecaches, x, pr, c, shp = cache
    grads = [None] * 20
    dz_pr = d_pr * pr * (1.0 - pr)
    dz_c = d_c * c * (1.0 - c)
    dx = dz_pr @ Wd[16].T + dz_c @ Wd[18].T
    if need_wgrad:
        grads[16], grads[17] = x.T @ dz_pr, dz_pr.sum(axis=0)
        grads[18], grads[19] = x.T @ dz_c, dz_c.sum(axis=0)
    egrads, d_in = encoder_bwd(cfg, Wd[0:16], ecaches, [None, None, None, dx.reshape(shp)],
                               need_wgrad=need_wgrad, need_input_grad=need_input_grad)
    grads[0:16] = egrads
    return grads, d_in


def adjuster_fwd(cfg, W, image, cond):
    """Adjuster.call model.py:125-136 (weights shared with D.encoder, G.decoder, G.conv)."""
    Wd, Wg, Wa = W["D"], W["G"], W["A"]
    outs, _ = encoder_fwd(cfg, Wd[0:16], image)
    u = cond @ Wa[0] + Wa[1]
    v = leaky(u, cfg.leaky_alpha)
    w2, nc = instnorm(v, Wa[2][0], Wa[3][0])  # 2-D input: reduces axis 1 (instance.py:107-112)
    w4 = w2.reshape(-1, cfg.init_dim, cfg.init_dim, cfg.conv_filter[0])
    add = outs[::-1]  # encoder_layers.reverse()  model.py:133
    xdec, dcaches = decoder_fwd(cfg, Wg[4:20], w4, add)
    pre = conv2d_transpose(_q(cfg, xdec), _q(cfg, Wg[20]), Wg[21], 1)
    img = np.tanh(pre)
    return img, (cond, u, nc, dcaches, img)


def adjuster_bwd_own(cfg, W, cache, d_img):
    """Gradient w.r.t. Adjuster.weights[16:20] only (eager_trainer.py:51,62,163)."""
    Wg, Wa = W["G"], W["A"]
    cond, u, nc, dcaches, img = cache
    dpre = d_img * (1.0 - img * img)
    dxdec = _q(cfg, conv_fwd(_q(cfg, dpre), _q(cfg, Wg[20]), 1))
    _, dw4 = decoder_bwd(cfg, Wg[4:20], dcaches, dxdec, need_wgrad=False)
    dv, dg, dbe = instnorm_bwd(nc, Wa[2][0], dw4.reshape(u.shape))
    du = leaky_bwd(u, dv, cfg.leaky_alpha)
    return [cond.T @ du, du.sum(axis=0), np.array([dg]), np.array([dbe])]


# --------------------------------------------------------------------------
# tf.compat.v1.train.AdamOptimizer   eager_trainer.py:28-30
# lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v updates; w -= lr_t*m/(sqrt(v)+eps); eps=1e-8.
# One (beta1_power, beta2_power) pair per OPTIMIZER, advanced on every
# apply_gradients call, whatever subset of variables it is given.
# --------------------------------------------------------------------------
ADAM_EPS = 1e-8


@dataclass
class AdamState:
    lr: float
    b1: float
    b2: float
    n: int
    b1p: float = field(init=False)
    b2p: float = field(init=False)
    m: List[Optional[np.ndarray]] = field(init=False)
    v: List[Optional[np.ndarray]] = field(init=False)

    def __post_init__(self):
        self.b1p, self.b2p = self.b1, self.b2
        self.m = [None] * self.n
        self.v = [None] * self.n

    def apply(self, weights: List[np.ndarray], idx: Sequence[int], grads: Sequence[np.ndarray]):
        lr_t = self.lr * math.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        for i, g in zip(idx, grads):
            if self.m[i] is None:
                self.m[i] = np.zeros_like(weights[i])
                self.v[i] = np.zeros_like(weights[i])
            self.m[i] = self.b1 * self.m[i] + (1.0 - self.b1) * g
            self.v[i] = self.b2 * self.v[i] + (1.0 - self.b2) * g * g
            weights[i] = weights[i] - lr_t * self.m[i] / (np.sqrt(self.v[i]) + ADAM_EPS)
        self.b1p *= self.b1
        self.b2p *= self.b2


@dataclass
class TrainState:
    cfg: Cfg
    W: Dict[str, List[np.ndarray]]
    opt: Dict[str, AdamState] = field(init=False)

    def __post_init__(self):
        c = self.cfg
        self.opt = {"G": AdamState(c.lr, c.beta_1, c.beta_2, 22),
                    "D": AdamState(c.lr, c.beta_1, c.beta_2, 20),
                    "A": AdamState(c.lr, 0.9, 0.999, 4)}  # eager_trainer.py:30 defaults


def step_gradients(cfg: Cfg, W, batch_no: int, inp: Dict[str, np.ndarray], fake_override=None, adj_override=None):
    """fake_override / adj_override (bf16-emulation tests only): images produced by the implementation under test,
    used downstream INSTEAD of this oracle's own generator / adjuster outputs (which are still computed, returned as
    fake_image_own / adj_image_own, and provide the caches of the generator / adjuster backward).  With bf16 rounding
    inside every layer, a 1e-5 difference between two implementations' images flips a fraction of the roundings in
    everything computed from them and decorrelates the two discriminator passes at the 1e-2 level; handing over the
    image at the model boundary keeps the comparison of each model's gradients tight.

    The arithmetic of eager_trainer.py:133-163 up to (not including) the optimizer
    applies.  `inp` holds real_image_1, real_cond_1, real_image_2, real_cond_2,
    noise, new_image (the RNG-dependent step inputs are inputs: SURVEY.md a17).
    Returns dict with fake_image, adj_image, losses and the three FULL gradient lists
    (the trainer then keeps the subset chosen by train_weight_indices)."""
    global _SCALE_LOG, _SCALE_TAG
    img1, c1, img2, c2 = inp["real_image_1"], inp["real_cond_1"], inp["real_image_2"], inp["real_cond_2"]
    noise, new_image = inp["noise"], inp["new_image"]
    out = {}
    _SCALE_LOG, _SCALE_TAG = {}, "fwd"
    try:
        return _step_gradients(cfg, W, batch_no, inp, fake_override, adj_override, out, img1, c1, img2, c2, noise, new_image)
    finally:
        _SCALE_LOG = None


def _scalar_scales(log):
    """tag -> per-call ((l1, l2) of dgamma, (l1, l2) of dbeta), calls in level order 4, 3, 2, 1 (+ the dense norm last for G / A)
    -> {model: {weight index: (l1, l2)}}; D's two passes add (the bound of a sum of two sums)."""
    sc = {"D": {}, "G": {}, "A": {}}
    for tag in ("D_real", "D_fake"):
        for j, (g, b) in enumerate(log.get(tag, [])):
            i = 3 - j
            for idx, v in ((4 * i + 2, g), (4 * i + 3, b)):
                o = sc["D"].get(idx, (0.0, 0.0))
                sc["D"][idx] = (o[0] + v[0], math.hypot(o[1], v[1]))
    recs = log.get("G", [])
    for j, (g, b) in enumerate(recs[:4]):
        i = 3 - j
        sc["G"][4 + 4 * i + 2], sc["G"][4 + 4 * i + 3] = g, b
    if len(recs) == 5:
        sc["G"][2], sc["G"][3] = recs[4]
    recs = log.get("A", [])
    if len(recs) == 5:
        sc["A"][2], sc["A"][3] = recs[4]
    return sc


def _step_gradients(cfg, W, batch_no, inp, fake_override, adj_override, out, img1, c1, img2, c2, noise, new_image):
    global _SCALE_TAG
    fake, gcache = generator_fwd(cfg, W["G"], noise, c2)
    out["fake_image_own"] = fake
    if fake_override is not None:
        fake = np.asarray(fake_override, F64)
        gcache = gcache[:-1] + (fake,)
    (real_pr, real_c), rcache = discriminator_fwd(cfg, W["D"], new_image)
    (fake_pr, fake_c), fcache = discriminator_fwd(cfg, W["D"], fake)
    # eager_trainer.py:85-91
    disc_loss = (2.0 * bce_mean(c1, real_c) + bce_mean(soft(1.0), real_pr) + bce_mean(soft(0.0), fake_pr))
    # eager_trainer.py:93-96
    gen_loss = (bce_mean(soft(1.0), fake_pr) + bce_mean(c2, fake_c) + cfg.l1_lambda * l1_mean(img2, fake))
    # disc tape (eager_trainer.py:145)
    _SCALE_TAG = "D_real"
    gr, _ = discriminator_bwd(cfg, W["D"], rcache, bce_mean_bwd(soft(1.0), real_pr),
                              2.0 * bce_mean_bwd(c1, real_c))
    _SCALE_TAG = "D_fake"
    gf, _ = discriminator_bwd(cfg, W["D"], fcache, bce_mean_bwd(soft(0.0), fake_pr), np.zeros_like(fake_c))
    _SCALE_TAG = "D_gen"
    dD = [a + b for a, b in zip(gr, gf)]
    # gen tape (eager_trainer.py:149)
    _, d_fake = discriminator_bwd(cfg, W["D"], fcache, bce_mean_bwd(soft(1.0), fake_pr),
                                  bce_mean_bwd(c2, fake_c), need_wgrad=False, need_input_grad=True)
    d_fake = d_fake + cfg.l1_lambda * l1_mean_bwd_b(img2, fake)
    _SCALE_TAG = "G"
    dG = generator_bwd(cfg, W["G"], gcache, d_fake)
    _SCALE_TAG = "A_disc"
    out.update(fake_image=fake, gen_loss=gen_loss, disc_loss=disc_loss, dD=dD, dG=dG,
               real_pr=real_pr, real_c=real_c, fake_pr=fake_pr, fake_c=fake_c,
               adj_image=None, adj_loss=None, dA=None)
    if cfg.train_adj and batch_no > 10:  # eager_trainer.py:152-163
        adj_in_cond = (np.concatenate([c2, c1], 0) + 1.0) * 0.5
        adj_t_cond = np.concatenate([c2, c1], 0)
        adj_in_img = np.concatenate([img1, fake], 0)
        adj_t_img = np.concatenate([img2, img1], 0)
        adj_img, acache = adjuster_fwd(cfg, W, adj_in_img, adj_in_cond)
        out["adj_image_own"] = adj_img
        if adj_override is not None:
            adj_img = np.asarray(adj_override, F64)
            acache = acache[:-1] + (adj_img,)
        (adj_pr, adj_c), dcache = discriminator_fwd(cfg, W["D"], adj_img)
        adj_loss = (bce_mean(soft(1.0), adj_pr) + bce_mean(adj_t_cond, adj_c)
                    + cfg.l1_lambda * l1_mean(adj_t_img, adj_img))  # eager_trainer.py:98-102
        _, d_adj = discriminator_bwd(cfg, W["D"], dcache, bce_mean_bwd(soft(1.0), adj_pr),
                                     bce_mean_bwd(adj_t_cond, adj_c), need_wgrad=False, need_input_grad=True)
        d_adj = d_adj + cfg.l1_lambda * l1_mean_bwd_b(adj_t_img, adj_img)
        _SCALE_TAG = "A"
        out.update(adj_image=adj_img, adj_loss=adj_loss, dA=adjuster_bwd_own(cfg, W, acache, d_adj))
    out["scalar_scale"] = _scalar_scales(_SCALE_LOG)
    return out


def train_step(state: TrainState, batch_no: int, inp: Dict[str, np.ndarray]):
    """eager_trainer.py:115-169: gradients on the pre-update snapshot, D-clip, then
    optimizer applies in the order Adjuster, Discriminator, Generator."""
    cfg, W = state.cfg, state.W
    out = step_gradients(cfg, W, batch_no, inp)
    iD = train_weight_indices(cfg, "D", batch_no)
    iG = train_weight_indices(cfg, "G", batch_no)
    gD = [out["dD"][i] for i in iD]
    if cfg.use_clip:  # eager_trainer.py:146-148
        gD = [np.clip(g, -cfg.clip_range, cfg.clip_range) for g in gD]
    gG = [out["dG"][i] for i in iG]
    if out["dA"] is not None:
        iA = train_weight_indices(cfg, "A", batch_no)
        state.opt["A"].apply(W["A"], iA, [out["dA"][i] for i in iA])
    state.opt["D"].apply(W["D"], iD, gD)
    state.opt["G"].apply(W["G"], iG, gG)
    return out


def make_inputs(cfg: Cfg, B: int, seed: int = 1234, pm_one: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic step inputs of SURVEY.md §8d: images U(-1,1), conds soft(+-1) (the
    CelebA convention through dataset.py:33) or soft({0,1}), noise N(0,1)."""
    rng = np.random.default_rng(seed)
    H = cfg.image_dim
    shp = (B, H, H, cfg.image_channel)

    def cond():
        bits = rng.integers(0, 2, size=(B, cfg.cond_dim)).astype(F64)
        return soft(2.0 * bits - 1.0) if pm_one else soft(bits)

    return {"real_image_1": rng.uniform(-1, 1, shp), "real_cond_1": cond(),
            "real_image_2": rng.uniform(-1, 1, shp), "real_cond_2": cond(),
            "noise": rng.standard_normal((B, cfg.noise_dim)),
            "new_image": rng.uniform(-1, 1, shp)}
