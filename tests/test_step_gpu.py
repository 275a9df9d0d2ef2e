"""Whole-step parity on the GPU: EagerTrainer.train_step_from_inputs (HIP kernels through the C ABI) vs
the fp64 numpy oracle on the same weights and inputs, incl. partition steps, the Adjuster branch (b > 10),
D-clip and the three TF-v1 Adam applies; plus the committed golden fixture.

Stated tolerances (parity to TensorFlow itself is UNPINNED, see oracle/np_oracle.py):
  f32 MFMA path : images 2e-5 abs, losses 2e-5 rel, gradients 2e-4 of max-abs per tensor
  bf16 MFMA path: images 3e-2 abs, losses 2e-2 rel, gradients 8e-2 of max-abs per tensor
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

TOLS = {"f32": dict(img=2e-5, loss=2e-5, grad=2e-4), "bf16": dict(img=3e-2, loss=2e-2, grad=8e-2)}
GOLD = os.path.join(os.path.dirname(__file__), "golden", "step_small.npz")


def make_args(cfg: O.Cfg, mfma_dtype="f32"):
    d = {k: getattr(cfg, k) for k in ("batch_size", "image_channel", "noise_dim", "init_dim", "conv_filter",
                                      "kernel_size", "leaky_alpha", "l1_lambda", "lr", "beta_1", "beta_2", "use_clip",
                                      "clip_range", "use_partition", "partition_interval", "train_adj", "cond_dim")}
    return SimpleNamespace(**d, mfma_dtype=mfma_dtype, device="cuda", seed=0, use_gp=False, no_io=True, dropout_rate=0.5)


def build(cfg, W, mfma_dtype):
    from littlegan_amd.eager_trainer import EagerTrainer
    from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator
    args = make_args(cfg, mfma_dtype)
    decoder, encoder = Decoder(args), Encoder(args)
    g = Generator(args, decoder)
    d = Discriminator(args, encoder)
    a = Adjuster(args, d, g)
    tr = EagerTrainer(args, g, d, a, None)
    with torch.no_grad():
        for dst, src in ((g.weights, W["G"]), (d.weights, W["D"]), (a.weights[16:20], W["A"])):
            assert len(dst) == len(src)
            for t, w in zip(dst, src):
                t.copy_(torch.tensor(w, dtype=torch.float32).view(t.shape))
    tr.store.bump()
    return tr


def dev_inputs(inp):
    return {k: torch.tensor(v, dtype=torch.float32, device="cuda").contiguous() for k, v in inp.items()}


def grads_of(tr, m):
    out = []
    for (s, e), name in zip(tr.store.ranges[m], tr.store.names(m)):
        out.append(tr.store.grad[s:e].detach().cpu().double().numpy())
    return out


def f32_round(d):
    return {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in d.items()}


def perturbed(cfg, seed):
    W = O.init_weights(cfg, seed)
    rng = np.random.default_rng(seed + 100)
    for m in W:
        for i, w in enumerate(W[m]):
            if w.ndim == 1:
                W[m][i] = w + 0.1 * rng.standard_normal(w.shape)
    return {m: [w.astype(np.float32).astype(np.float64) for w in ws] for m, ws in W.items()}


@pytest.mark.parametrize("mfma", ["f32", "bf16"])
def test_step_matches_oracle_small(mfma):
    tol = TOLS[mfma]
    cfg = O.Cfg(init_dim=2, conv_filter=(64, 32, 32, 32, 32), cond_dim=5, noise_dim=11, batch_size=3)
    W = perturbed(cfg, 1)
    st = O.TrainState(cfg, {m: [w.copy() for w in ws] for m, ws in W.items()})
    tr = build(cfg, W, mfma)
    for b in (4, 5, 11, 15):
        inp = f32_round(O.make_inputs(cfg, cfg.batch_size, seed=50 + b))
        W_before = {m: [w.copy() for w in ws] for m, ws in st.W.items()}
        ref = O.train_step(st, b, inp)
        fake, adj, lg, ld, la = tr.train_step_from_inputs(b, dev_inputs(inp))
        assert np.abs(fake.cpu().numpy() - ref["fake_image"]).max() < tol["img"]
        assert abs(lg.item() - ref["gen_loss"]) < tol["loss"] * abs(ref["gen_loss"])
        assert abs(ld.item() - ref["disc_loss"]) < tol["loss"] * abs(ref["disc_loss"])
        sets = [("D", "dD"), ("G", "dG")]
        if b > 10:
            assert np.abs(adj.cpu().numpy() - ref["adj_image"]).max() < tol["img"]
            assert abs(la.item() - ref["adj_loss"]) < tol["loss"] * abs(ref["adj_loss"])
            sets.append(("A", "dA"))
        else:
            assert adj is None and la is None
        for m, key in sets:
            for i, (got, exp) in enumerate(zip(grads_of(tr, m), ref[key])):
                exp = np.asarray(exp, np.float64).ravel()
                err = np.abs(got[:exp.size] - exp).max()
                assert err <= tol["grad"] * (np.abs(exp).max() + 1e-12), (b, m, i, err, np.abs(exp).max())
        if mfma == "f32":
            # weights after Adam: a per-element update is at most ~lr; near-zero gradients may flip sign
            for m, dst in (("G", tr.generator.weights), ("D", tr.discriminator.weights), ("A", tr.adjuster.weights[16:20])):
                for t, w, w0 in zip(dst, st.W[m], W_before[m]):
                    d = np.abs(t.detach().cpu().double().numpy().ravel() - w.ravel())
                    assert d.max() <= 2.2 * cfg.lr * 3.2 and d.mean() <= 0.02 * cfg.lr, (b, m, d.max(), d.mean())
            # re-sync so that later steps compare like with like
            with torch.no_grad():
                for m, dst in (("G", tr.generator.weights), ("D", tr.discriminator.weights), ("A", tr.adjuster.weights[16:20])):
                    for t, w in zip(dst, st.W[m]):
                        t.copy_(torch.tensor(w, dtype=torch.float32).view(t.shape))
        else:
            with torch.no_grad():
                for m, dst in (("G", tr.generator.weights), ("D", tr.discriminator.weights), ("A", tr.adjuster.weights[16:20])):
                    for t, w in zip(dst, st.W[m]):
                        t.copy_(torch.tensor(w, dtype=torch.float32).view(t.shape))
        tr.store.bump()


def test_partition_and_beta_powers():
    """Only the scheduled group moves on a partition step, and every optimizer's beta powers advance once per apply."""
    cfg = O.Cfg(init_dim=2, conv_filter=(32, 32, 32, 32, 32), cond_dim=3, noise_dim=5, batch_size=2)
    W = perturbed(cfg, 3)
    tr = build(cfg, W, "f32")
    before = tr.store.flat.clone()
    inp = dev_inputs(f32_round(O.make_inputs(cfg, 2, seed=1)))
    tr.train_step_from_inputs(5, inp)  # b=5: G group 1 = w[4:8], D group 1 = w[12:16]; no Adjuster (b <= 10)
    changed = (tr.store.flat != before)
    for m, (lo, hi) in (("G", (4, 8)), ("D", (12, 16))):
        for i, (s, e) in enumerate(tr.store.ranges[m]):
            if lo <= i < hi:
                assert changed[s:e].any(), (m, i)
            else:
                assert not changed[s:e].any(), (m, i)
    sA, eA = tr.store.model_range("A")
    assert not changed[sA:eA].any()
    assert tr.opt_state["G"].tolist() == pytest.approx([0.25, 0.81]) and tr.opt_state["A"].tolist() == pytest.approx([0.9, 0.999])


def test_golden_fixture_f32():
    from tests.golden.make_golden import CFG, STEPS
    g = np.load(GOLD)
    cfg = O.Cfg(**CFG)
    shp = O.weight_shapes(cfg)
    W = {m: [g[f"W0_{m}_{i}"].astype(np.float64) for i in range(len(shp[m]))] for m in "GDA"}
    tr = build(cfg, W, "f32")
    for b in STEPS:
        inp = {k: g[f"in{b}_{k}"] for k in ("real_image_1", "real_cond_1", "real_image_2", "real_cond_2", "noise", "new_image")}
        fake, adj, lg, ld, la = tr.train_step_from_inputs(b, dev_inputs(inp))
        # weights drift from the oracle's by O(lr) per step (Adam sign sensitivity), so later steps get looser bounds
        k = 1 + STEPS.index(b)
        assert np.abs(fake.cpu().numpy() - g[f"out{b}_fake_image"]).max() < 2e-5 + 2e-3 * (k - 1)
        assert abs(lg.item() - float(g[f"out{b}_gen_loss"])) < (2e-5 + 1e-3 * (k - 1)) * abs(float(g[f"out{b}_gen_loss"]))
        assert abs(ld.item() - float(g[f"out{b}_disc_loss"])) < (2e-5 + 1e-3 * (k - 1)) * abs(float(g[f"out{b}_disc_loss"]))
    for m, dst in (("G", tr.generator.weights), ("D", tr.discriminator.weights), ("A", tr.adjuster.weights[16:20])):
        for i, t in enumerate(dst):
            d = np.abs(t.detach().cpu().numpy().ravel() - g[f"W4_{m}_{i}"].ravel())
            assert d.mean() < 0.05 * cfg.lr * len(STEPS), (m, i, d.mean())


@pytest.mark.parametrize("mfma", ["f32", "bf16"])
def test_step_full_channels_one_step(mfma):
    """Reference channel widths (384..32), 64x64 images, B=2: exercises every tile configuration."""
    tol = TOLS[mfma]
    cfg = O.Cfg(init_dim=4, cond_dim=40, batch_size=2)
    W = perturbed(cfg, 7)
    tr = build(cfg, W, mfma)
    inp = f32_round(O.make_inputs(cfg, 2, seed=9))
    ref = O.step_gradients(cfg, W, 11, inp)
    fake, adj, lg, ld, la = tr.train_step_from_inputs(11, dev_inputs(inp))
    assert np.abs(fake.cpu().numpy() - ref["fake_image"]).max() < tol["img"]
    assert np.abs(adj.cpu().numpy() - ref["adj_image"]).max() < tol["img"]
    for got, key in ((lg, "gen_loss"), (ld, "disc_loss"), (la, "adj_loss")):
        assert abs(got.item() - ref[key]) < tol["loss"] * abs(ref[key])
    for m, key in (("D", "dD"), ("G", "dG"), ("A", "dA")):
        for i, (got, exp) in enumerate(zip(grads_of(tr, m), ref[key])):
            exp = np.asarray(exp, np.float64).ravel()
            err = np.abs(got[:exp.size] - exp).max()
            assert err <= tol["grad"] * (np.abs(exp).max() + 1e-12), (m, i, err, np.abs(exp).max())
