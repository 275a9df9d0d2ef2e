"""Input side of the step (eager_trainer.py:125-131): the counter-based generator and the image
augmentation.  CPU part: the oracle's Philox4x32-10 against the published Random123 known-answer vectors, and the
algebra of the restated image ops.  GPU part: the device kernels against that oracle, bit-exact for the generator."""
import numpy as np
import pytest
import torch

from oracle import input_oracle as I


def test_philox_known_answer_vectors():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert I.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert I.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert I.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                           [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_image_ops_algebra():
    rng = np.random.default_rng(0)
    img = rng.uniform(-1, 1, (3, 6, 5, 3))
    # hue rotation: identity at 0, inverse at -d, a full turn is the identity, min and max of every pixel preserved
    assert np.allclose(I.hue_rotate(img, 0.0), img)
    h = I.hue_rotate(img, 0.03)
    assert np.allclose(I.hue_rotate(h, -0.03), img, atol=1e-12)
    assert np.allclose(I.hue_rotate(img, 1.0), img, atol=1e-12)
    assert np.allclose(h.max(-1), img.max(-1)) and np.allclose(h.min(-1), img.min(-1))
    # a third of a turn permutes the channels: (r,g,b) -> (b,r,g)
    assert np.allclose(I.hue_rotate(img, 1.0 / 3.0), img[..., [2, 0, 1]], atol=1e-12)
    # contrast keeps the per-image per-channel mean, brightness shifts it, flip mirrors x
    out = I.augment(img, [True, False, True], 0.01, 0.8, 0.0)
    assert np.allclose(out.mean((1, 2)), img.mean((1, 2)) + 0.01)
    assert np.allclose(out[1], (img[1] + 0.01 - (img[1].mean((0, 1)) + 0.01)) * 0.8 + img[1].mean((0, 1)) + 0.01)
    assert np.allclose(I.augment(img, [True] * 3, 0.0, 1.0, 0.0), img[:, :, ::-1])


@pytest.mark.gpu
def test_device_generator_is_the_oracle_generator():
    from littlegan_amd import ops
    seed, off = 0x0123456789ABCDEF, (7 << 40) + 5
    got = ops.philox4x32(300, seed, off).cpu().numpy().view(np.uint32).reshape(300, 4)
    assert np.array_equal(got, I.philox_blocks(300, seed, off))           # bit-exact
    z = ops.randn((1001,), seed, off, mean=0.5, std=2.0).cpu().numpy()
    exp = 0.5 + 2.0 * I.normals(251, seed, off).reshape(-1)[:1001]
    assert np.abs(z - exp).max() < 2e-5
    # the same counter window gives the same numbers, a different one different numbers; moments of a large draw
    a = ops.randn((1 << 20,), 3, 1 << 40)
    assert torch.equal(a, ops.randn((1 << 20,), 3, 1 << 40)) and not torch.equal(a, ops.randn((1 << 20,), 3, 2 << 40))
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1.0) < 5e-3
    assert abs((a ** 3).mean().item()) < 2e-2 and abs((a ** 4).mean().item() - 3.0) < 5e-2


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 8, 10), (2, 64, 64), (5, 17, 9)])
def test_device_augmentation_matches_the_oracle(shape):
    from littlegan_amd import ops
    B, H, W = shape
    rng = np.random.default_rng(B * 100 + H)
    img = rng.uniform(-1, 1, (B, H, W, 3)).astype(np.float32)
    img[0, 0, 0] = 0.25            # a grey pixel: hue undefined, must pass through
    flip = rng.random(B) < 0.5
    db, cf, dh = 0.013, 0.81, -0.021
    x = torch.tensor(img, device="cuda")
    f = torch.tensor(flip.astype(np.uint8), device="cuda")
    out = ops.augment(x, f, db, cf, dh, 0.0, 1, 0).cpu().numpy()
    exp = I.augment(img.astype(np.float64), flip, db, cf, dh)
    assert np.abs(out - exp).max() < 3e-6
    # with the noise term: out - deterministic part = 0.02 * the oracle's normals of block (offset + pixel)
    seed, off = 99, 3 << 38
    outn = ops.augment(x, f, db, cf, dh, 0.02, seed, off).cpu().numpy()
    nz = I.normals(B * H * W, seed, off)[:, :3].reshape(B, H, W, 3)
    assert np.abs(outn - (exp + 0.02 * nz)).max() < 5e-6
    # no flip mask and no hue shift are valid calls
    out2 = ops.augment(x, None, 0.0, 1.0, 0.0, 0.0, 0, 0)
    assert torch.allclose(out2, x, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 6, 256])
def test_device_side_draws_match_the_oracle(B):
    """lg_augment_drawn: flip / brightness / contrast / hue drawn on the device from the Philox window (no host round
    trip) == the oracle's transform fed with the oracle's reading of the same window."""
    from littlegan_amd import ops
    H = W = 16
    rng = np.random.default_rng(B)
    img = rng.uniform(-1, 1, (B, H, W, 3)).astype(np.float32)
    seed, doff, noff = (5 << 20) ^ 1, (9 << 40) + (1 << 39), (9 << 40) + (1 << 38)
    db, cf, dh, flip = I.step_draws(B, seed, doff)
    assert abs(db) <= 0.02 and 0.75 <= cf <= 1.003 and abs(dh) <= 0.03
    x = torch.tensor(img, device="cuda")
    out = ops.augment_drawn(x, 0.02, 0.75, 1.003, 0.03, 0.0, seed, doff, noff).cpu().numpy()
    exp = I.augment(img.astype(np.float64), flip, db, cf, dh)
    assert np.abs(out - exp).max() < 5e-6
    outn = ops.augment_drawn(x, 0.02, 0.75, 1.003, 0.03, 0.02, seed, doff, noff).cpu().numpy()
    nz = I.normals(B * H * W, seed, noff)[:, :3].reshape(B, H, W, 3)
    assert np.abs(outn - (exp + 0.02 * nz)).max() < 8e-6
    if B == 256:
        assert 90 < int(flip.sum()) < 166   # a fair coin per image


@pytest.mark.gpu
def test_trainer_draws_reproducible_step_inputs():
    from test_step_gpu import build, perturbed
    from oracle import np_oracle as O
    cfg = O.Cfg(init_dim=2, conv_filter=(32, 32, 32, 32, 32), cond_dim=3, noise_dim=5, batch_size=4)
    img = torch.rand(4, 32, 32, 3, device="cuda") * 2 - 1
    tr1, tr2 = build(cfg, perturbed(cfg, 3), "f32"), build(cfg, perturbed(cfg, 3), "f32")
    n1, a1 = tr1.draw_step_inputs(img)
    n2, a2 = tr2.draw_step_inputs(img)
    assert torch.equal(n1, n2) and torch.equal(a1, a2)          # same (seed, rank, step) -> same draws
    n3, a3 = tr1.draw_step_inputs(img)
    assert not torch.equal(n1, n3) and not torch.equal(a1, a3)  # next step -> new window
    assert n1.shape == (4, 5) and a1.shape == img.shape
    assert (a1 - img).abs().max() < 2.5                          # flipped / shifted / noised, still image-scaled
