"""CPU tests of the oracle: the known-answer tests derivable from the reference
source alone, numpy-manual-backward vs torch-autograd cross-check,
and the committed golden fixture.  Parity to TensorFlow itself is UNPINNED."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle import torch_oracle as T

GOLD = os.path.join(os.path.dirname(__file__), "golden", "step_small.npz")


def test_soft_and_rescale_known_answers():
    # utils.py:47-56
    assert O.soft(1.0) == pytest.approx(0.98) and O.soft(0.0) == pytest.approx(0.02)
    assert O.soft(-1.0) == pytest.approx(-0.94)
    assert O.data_rescale(0) == -1.0 and O.data_rescale(255) == 1.0
    assert O.inverse_rescale(-1.0) == 0.0 and O.inverse_rescale(1.0) == 255.0


def test_same_padding_split():
    # even size, k5, s2: pad_total 3 -> 1 before, 2 after ; s1: 2/2
    assert O.same_pads(128, 5, 2) == (64, 1, 2)
    assert O.same_pads(8, 5, 1) == (8, 2, 2)


def test_conv_delta_kernels_pin_alignment():
    """A delta kernel at tap (ky,kx) must shift by (ky-1, kx-1) for the stride-2 conv
    and the transposed conv must place x[i] at o = 2*i + ky - 1."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 8, 8, 1))
    for ky, kx in [(0, 0), (1, 1), (4, 4), (2, 3)]:
        w = np.zeros((5, 5, 1, 1))
        w[ky, kx, 0, 0] = 1.0
        y = O.conv2d(x, w, np.zeros(1), 2)
        for oy in range(4):
            for ox in range(4):
                iy, ix = 2 * oy + ky - 1, 2 * ox + kx - 1
                exp = x[0, iy, ix, 0] if 0 <= iy < 8 and 0 <= ix < 8 else 0.0
                assert y[0, oy, ox, 0] == exp
        xt = rng.standard_normal((1, 4, 4, 1))
        yt = O.conv2d_transpose(xt, w, np.zeros(1), 2)
        exp = np.zeros((8, 8))
        for iy in range(4):
            for ix in range(4):
                oy, ox = 2 * iy + ky - 1, 2 * ix + kx - 1
                if 0 <= oy < 8 and 0 <= ox < 8:
                    exp[oy, ox] = xt[0, iy, ix, 0]
        assert np.array_equal(yt[0, :, :, 0], exp)


def test_instnorm_constant_sample_is_beta():
    # instance.py:114-127: x-mean == 0 exactly -> y == beta
    x = np.full((2, 4, 4, 3), 3.25)
    y, _ = O.instnorm(x, 1.7, -0.4)
    assert np.array_equal(y, np.full_like(x, -0.4))


def test_instnorm_eps_added_to_std():
    x = np.array([[1.0, -1.0, 1.0, -1.0]])  # std = 1
    y, _ = O.instnorm(x, 1.0, 0.0)
    assert np.allclose(y, x / 1.001, rtol=0, atol=1e-15)


def test_bce_known_answer():
    # BCE(t=0.98, p=0.5) = -log(0.5 + 1e-7)
    p = np.full((4, 1), 0.5)
    assert O.bce_mean(0.98, p) == pytest.approx(-math.log(0.5 + 1e-7), rel=1e-14)


def test_weight_counts_and_groups():
    cfg = O.Cfg(cond_dim=40)
    shp = O.weight_shapes(cfg)
    assert len(shp["G"]) == 22 and len(shp["D"]) == 20 and len(shp["A"]) == 4
    W = O.init_weights(cfg)
    assert len(O.adjuster_weights(W)) == 38
    aw = O.adjuster_weights(W)
    assert all(a is b for a, b in zip(aw[16:20], W["A"]))  # eager_trainer.py:51
    n = lambda m: sum(int(np.prod(s)) for _, s in shp[m])
    # G 6.83 M, D 4.49 M, A-own 1.01 M  (c=40, 128^2)
    assert round(n("G") / 1e6, 2) == 6.83 and round(n("D") / 1e6, 2) == 4.49 and round(n("A") / 1e6, 2) == 1.01
    # kernel layouts: conv HWIO, convT HWOI
    assert shp["D"][0][1] == (5, 5, 3, 64) and shp["G"][4][1] == (5, 5, 256, 384)
    assert shp["G"][20][1] == (5, 5, 3, 32)


def test_partition_schedule():
    # eager_trainer.py:104-113 ; b=5 -> group 1, b=10 -> group 2, b=15 -> group 0
    cfg = O.Cfg()
    for b in range(1, 16):
        g, d, a = (O.train_weight_indices(cfg, m, b) for m in "GDA")
        if b % 5:
            assert g == list(range(22)) and d == list(range(20)) and a == list(range(4))
    assert O.train_weight_indices(cfg, "G", 5) == list(range(4, 8))
    assert O.train_weight_indices(cfg, "G", 10) == list(range(8, 22))
    assert O.train_weight_indices(cfg, "G", 15) == list(range(0, 4))
    assert O.train_weight_indices(cfg, "D", 5) == list(range(12, 16))
    assert O.train_weight_indices(cfg, "D", 10) == list(range(16, 20))
    assert O.train_weight_indices(cfg, "D", 15) == list(range(0, 12))
    assert O.train_weight_indices(cfg, "A", 5) == list(range(4))
    cfg.use_partition = False
    assert O.train_weight_indices(cfg, "G", 5) == list(range(22))


def test_adam_first_step_and_shared_beta_power():
    # first step: m = (1-b1) g, v = (1-b2) g^2, lr_t = lr*sqrt(1-b2)/(1-b1) -> dw = -lr*g/(|g| + eps*sqrt(1-b2))
    st = O.AdamState(5e-5, 0.5, 0.9, 2)
    w = [np.array([1.0, -2.0]), np.array([0.5])]
    g = np.array([0.3, -4.0])
    st.apply(w, [0], [g])
    exp = np.array([1.0, -2.0]) - 5e-5 * math.sqrt(0.1) / 0.5 * (0.5 * g) / (np.sqrt(0.1 * g * g) + 1e-8)
    assert np.allclose(w[0], exp, rtol=1e-15)
    assert np.allclose(w[0] - np.array([1.0, -2.0]), -5e-5 * np.sign(g), rtol=1e-6)
    # beta powers advanced although variable 1 was not in the subset; its slots untouched
    assert st.b1p == 0.25 and st.b2p == pytest.approx(0.81) and st.m[1] is None


def test_shapes_full_config():
    cfg = O.Cfg(init_dim=2, cond_dim=4, noise_dim=6, conv_filter=(16, 8, 8, 8, 8))
    W = O.init_weights(cfg)
    inp = O.make_inputs(cfg, 2)
    img, _ = O.generator_fwd(cfg, W["G"], inp["noise"], inp["real_cond_2"])
    assert img.shape == (2, 32, 32, 3)
    (pr, c), _ = O.discriminator_fwd(cfg, W["D"], img)
    assert pr.shape == (2, 1) and c.shape == (2, 4)
    adj, _ = O.adjuster_fwd(cfg, W, img, inp["real_cond_1"])
    assert adj.shape == img.shape
    assert set(np.unique(inp["real_cond_1"])) <= {O.soft(-1.0), O.soft(1.0)}


def _perturbed_weights(cfg, seed):
    W = O.init_weights(cfg, seed)
    rng = np.random.default_rng(seed + 100)
    for m in W:
        for i, w in enumerate(W[m]):
            if w.ndim == 1:
                W[m][i] = w + 0.1 * rng.standard_normal(w.shape)
    return W


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


@pytest.mark.parametrize("batch_no", [3, 11])
def test_manual_backward_matches_autograd(batch_no):
    """np_oracle's hand-written backward == torch autograd on an independent forward."""
    cfg = O.Cfg(init_dim=2, conv_filter=(32, 16, 16, 8, 8), cond_dim=5, noise_dim=11)
    W = _perturbed_weights(cfg, 1)
    inp = O.make_inputs(cfg, 3, 7)
    o = O.step_gradients(cfg, W, batch_no, inp)
    t = T.step_gradients(T.Net(cfg, W), batch_no, {k: torch.tensor(v) for k, v in inp.items()})
    for k in ("fake_image", "gen_loss", "disc_loss"):
        assert _rel(o[k], t[k].numpy()) < 1e-12
    for k in ("dD", "dG"):
        for a, b in zip(o[k], t[k]):
            assert _rel(a, b.numpy().reshape(a.shape)) < 1e-10
    if batch_no > 10:
        assert _rel(o["adj_image"], t["adj_image"].numpy()) < 1e-12
        assert _rel(o["adj_loss"], t["adj_loss"].numpy()) < 1e-12
        for a, b in zip(o["dA"], t["dA"]):
            assert _rel(a, b.numpy().reshape(a.shape)) < 1e-10
    else:
        assert o["adj_image"] is None and o["dA"] is None  # eager_trainer.py:152  b > 10


def test_torch_trainer_matches_numpy_trainer_over_steps():
    """Whole step incl. partition subsets, clip, Adam order; fp64 both sides."""
    cfg = O.Cfg(init_dim=2, conv_filter=(16, 8, 8, 8, 8), cond_dim=3, noise_dim=5)
    W = _perturbed_weights(cfg, 2)
    st = O.TrainState(cfg, {m: [w.copy() for w in ws] for m, ws in W.items()})
    tr = T.Trainer(cfg, W, dtype=torch.float64)
    for b in (9, 10, 11, 15):
        inp = O.make_inputs(cfg, 2, 50 + b)
        O.train_step(st, b, inp)
        tr.step(b, {k: torch.tensor(v) for k, v in inp.items()})
    for m in "GDA":
        for a, b_ in zip(st.W[m], tr.net.W[m]):
            assert _rel(a, b_.detach().numpy()) < 1e-10
    assert st.opt["A"].b1p == pytest.approx(0.9 ** 3)  # applied only on b = 11, 15
    assert st.opt["G"].b1p == pytest.approx(0.5 ** 5)


def test_golden_fixture_reproduces():
    from tests.golden.make_golden import CFG, STEPS
    g = np.load(GOLD)
    cfg = O.Cfg(**CFG)
    shp = O.weight_shapes(cfg)
    W = {m: [g[f"W0_{m}_{i}"].astype(np.float64) for i in range(len(shp[m]))] for m in "GDA"}
    st = O.TrainState(cfg, W)
    for b in STEPS:
        inp = {k: g[f"in{b}_{k}"].astype(np.float64) for k in
               ("real_image_1", "real_cond_1", "real_image_2", "real_cond_2", "noise", "new_image")}
        out = O.train_step(st, b, inp)
        assert out["gen_loss"] == pytest.approx(float(g[f"out{b}_gen_loss"]), rel=1e-12)
        assert out["disc_loss"] == pytest.approx(float(g[f"out{b}_disc_loss"]), rel=1e-12)
        assert _rel(out["fake_image"], g[f"out{b}_fake_image"]) < 1e-6  # stored as f32
        if b > 10:
            assert out["adj_loss"] == pytest.approx(float(g[f"out{b}_adj_loss"]), rel=1e-12)
    for m in "GDA":
        for i, w in enumerate(st.W[m]):
            assert _rel(w, g[f"W4_{m}_{i}"]) < 1e-6
