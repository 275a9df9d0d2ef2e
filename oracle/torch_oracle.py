"""torch-CPU autograd restatement of the LittleGAN training step.

TEST INFRASTRUCTURE ONLY (same rule as np_oracle.py): used (a) as an INDEPENDENT
cross-check of np_oracle's hand-written backward passes (autograd vs manual), and
(b) as the "CPU restatement, not TensorFlow" baseline timed by bench.py's
``cpu_baseline`` leg and by the gloo data-parallel tests.  PARITY UNPINNED — see
np_oracle.py's header.

Independent of np_oracle on purpose: convolutions go through torch's
conv2d / conv_transpose2d with manual TF-SAME pad / crop instead of the tap loops.
Weights use the reference (TF) layouts and ordering; tensors are NHWC at the API.
References are to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

IN_EPS = 1e-3
import numpy as _np
BCE_EPS = float(_np.float32(1e-7))  # float32 constants, see np_oracle.py
BCE_HI = float(_np.float32(1.0) - _np.float32(1e-7))
ADAM_EPS = 1e-8


def soft(x):  # utils.py:47-48
    return 0.96 * x + 0.02


def _same_pads(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv2d_same(x, w, b, s):
    """model.py:15 ; x NHWC, w HWIO."""
    pt, pb = _same_pads(x.shape[1], w.shape[0], s)
    pl, pr = _same_pads(x.shape[2], w.shape[1], s)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=s)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_same(x, w, b, s):
    """model.py:39-40,86-87 ; x NHWC, w HWOI [k,k,Co,Ci]; out side = s*in.
    Full transposed conv (o = s*i + k) cropped at pad_before of the forward conv whose
    input is the s*in-sized output."""
    k = w.shape[0]
    H, W = x.shape[1] * s, x.shape[2] * s
    pt, _ = _same_pads(H, k, s)
    pl, _ = _same_pads(W, k, s)
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, stride=s)
    y = full[:, :, pt:pt + H, pl:pl + W] + b.view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1)


def instnorm(x, gamma, beta):
    """instance.py:105-128, axis=None."""
    B = x.shape[0]
    xf = x.reshape(B, -1)
    mu = xf.mean(dim=1, keepdim=True)
    sd = ((xf - mu) ** 2).mean(dim=1, keepdim=True).sqrt() + IN_EPS
    return (((xf - mu) / sd) * gamma + beta).reshape(x.shape)


def bce_mean(t, p):
    """tf.keras.losses.binary_crossentropy (TF-1.15 backend form) + reduce_mean."""
    if not torch.is_tensor(t):
        t = torch.full_like(p, float(t))
    pc = torch.clamp(p, BCE_EPS, BCE_HI)
    l = -(t * torch.log(pc + BCE_EPS) + (1.0 - t) * torch.log(1.0 - pc + BCE_EPS))
    return l.mean(dim=-1).mean()


class Net:
    """Holds W = {'G': [22], 'D': [20], 'A': [4]} as leaf tensors (reference layouts)."""

    def __init__(self, cfg, W_np: Dict[str, List], dtype=torch.float64):
        self.cfg = cfg
        self.dtype = dtype
        self.W = {m: [torch.tensor(w, dtype=dtype).requires_grad_(True) for w in ws] for m, ws in W_np.items()}

    # model.py:18-27
    def encoder(self, x):
        We, a = self.W["D"], self.cfg.leaky_alpha
        outs = []
        for i in range(4):
            k, b, g, be = We[4 * i:4 * i + 4]
            x = F.leaky_relu(instnorm(conv2d_same(x, k, b, 2), g, be), a)
            outs.append(x)
        return outs

    # model.py:43-51
    def decoder(self, x, add):
        Wd, a = self.W["G"][4:20], self.cfg.leaky_alpha
        for i in range(4):
            k, b, g, be = Wd[4 * i:4 * i + 4]
            if add[i] is not None:
                x = x + add[i]
            x = F.leaky_relu(instnorm(conv2d_transpose_same(x, k, b, 2), g, be), a)
        return x

    # model.py:89-105
    def generator(self, noise, cond):
        Wg, c = self.W["G"], self.cfg
        x = torch.cat([noise, cond], dim=-1) @ Wg[0] + Wg[1]
        x = F.leaky_relu(x, c.leaky_alpha).reshape(-1, c.init_dim, c.init_dim, c.conv_filter[0])
        x = instnorm(x, Wg[2], Wg[3])
        x = self.decoder(x, [None] * 4)
        return torch.tanh(conv2d_transpose_same(x, Wg[20], Wg[21], 1))

    # model.py:65-73
    def discriminator(self, image):
        Wd = self.W["D"]
        x = self.encoder(image)[-1].reshape(image.shape[0], -1)
        return torch.sigmoid(x @ Wd[16] + Wd[17]), torch.sigmoid(x @ Wd[18] + Wd[19])

    # model.py:125-136
    def adjuster(self, image, cond):
        Wa, Wg, c = self.W["A"], self.W["G"], self.cfg
        enc = self.encoder(image)
        x = F.leaky_relu(cond @ Wa[0] + Wa[1], c.leaky_alpha)
        x = instnorm(x, Wa[2], Wa[3]).reshape(-1, c.init_dim, c.init_dim, c.conv_filter[0])
        x = self.decoder(x, enc[::-1])
        return torch.tanh(conv2d_transpose_same(x, Wg[20], Wg[21], 1))


def step_gradients(net: Net, batch_no: int, inp: Dict[str, torch.Tensor]):
    """eager_trainer.py:133-163 with autograd playing the GradientTapes."""
    cfg, W = net.cfg, net.W
    img1, c1, img2, c2 = inp["real_image_1"], inp["real_cond_1"], inp["real_image_2"], inp["real_cond_2"]
    fake = net.generator(inp["noise"], c2)
    real_pr, real_c = net.discriminator(inp["new_image"])
    fake_pr, fake_c = net.discriminator(fake)
    disc_loss = 2.0 * bce_mean(c1, real_c) + bce_mean(soft(1.0), real_pr) + bce_mean(soft(0.0), fake_pr)
    gen_loss = (bce_mean(soft(1.0), fake_pr) + bce_mean(c2, fake_c)
                + cfg.l1_lambda * (img2 - fake).abs().mean())
    dD = torch.autograd.grad(disc_loss, W["D"], retain_graph=True)
    dG = torch.autograd.grad(gen_loss, W["G"], retain_graph=False)
    out = dict(fake_image=fake.detach(), gen_loss=gen_loss.detach(), disc_loss=disc_loss.detach(),
               dD=list(dD), dG=list(dG), adj_image=None, adj_loss=None, dA=None)
    if cfg.train_adj and batch_no > 10:
        fk = fake.detach()
        adj_in_cond = (torch.cat([c2, c1], 0) + 1.0) * 0.5
        adj_t_cond = torch.cat([c2, c1], 0)
        adj_img = net.adjuster(torch.cat([img1, fk], 0), adj_in_cond)
        adj_pr, adj_c = net.discriminator(adj_img)
        adj_loss = (bce_mean(soft(1.0), adj_pr) + bce_mean(adj_t_cond, adj_c)
                    + cfg.l1_lambda * (torch.cat([img2, img1], 0) - adj_img).abs().mean())
        dA = torch.autograd.grad(adj_loss, W["A"])
        out.update(adj_image=adj_img.detach(), adj_loss=adj_loss.detach(), dA=list(dA))
    return out


class Adam:
    """tf.compat.v1.train.AdamOptimizer (eager_trainer.py:28-30); shared beta powers."""

    def __init__(self, lr, b1, b2, weights):
        self.lr, self.b1, self.b2 = lr, b1, b2
        self.b1p, self.b2p = b1, b2
        self.m = [torch.zeros_like(w) for w in weights]
        self.v = [torch.zeros_like(w) for w in weights]

    @torch.no_grad()
    def apply(self, weights, idx, grads):
        lr_t = self.lr * math.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        for i, g in zip(idx, grads):
            self.m[i].mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            self.v[i].mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            weights[i].sub_(lr_t * self.m[i] / (self.v[i].sqrt() + ADAM_EPS))
        self.b1p *= self.b1
        self.b2p *= self.b2


class Trainer:
    """Whole step incl. partition schedule, D-clip and the A->D->G apply order
    (eager_trainer.py:104-169).  `allreduce` (optional) is called on each gradient
    list before clipping — the hook the gloo data-parallel tests use."""

    def __init__(self, cfg, W_np, dtype=torch.float32, allreduce=None):
        from oracle.np_oracle import train_weight_indices  # host-side schedule, same file family
        self._idx = train_weight_indices
        self.cfg = cfg
        self.net = Net(cfg, W_np, dtype)
        W = self.net.W
        self.opt = {"G": Adam(cfg.lr, cfg.beta_1, cfg.beta_2, W["G"]),
                    "D": Adam(cfg.lr, cfg.beta_1, cfg.beta_2, W["D"]),
                    "A": Adam(cfg.lr, 0.9, 0.999, W["A"])}
        self.allreduce = allreduce

    def step(self, batch_no, inp):
        cfg, W = self.cfg, self.net.W
        out = step_gradients(self.net, batch_no, inp)
        sel = {}
        for m, key in (("A", "dA"), ("D", "dD"), ("G", "dG")):
            if out[key] is None:
                continue
            idx = self._idx(cfg, m, batch_no)
            g = [out[key][i] for i in idx]
            if self.allreduce is not None:
                g = self.allreduce(g)
            if m == "D" and cfg.use_clip:
                g = [x.clamp(-cfg.clip_range, cfg.clip_range) for x in g]
            sel[m] = (idx, g)
        for m in ("A", "D", "G"):
            if m in sel:
                self.opt[m].apply(W[m], *sel[m])
        return out
