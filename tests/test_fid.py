"""FID arithmetic (fid.py:112-163,185-188) pinned by the known answers derivable from the source:
identical statistics -> 0, the closed form for diagonal covariances, and np.cov for the statistics."""
import warnings

import numpy as np
import pytest
import torch

from littlegan_amd.fid import activation_statistics, fid_from_activations, frechet_distance


def test_identical_statistics_give_zero():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((50, 8))
    s = np.cov(a, rowvar=False)
    assert abs(frechet_distance(a.mean(0), s, a.mean(0), s)) < 1e-9


def test_diagonal_covariances_closed_form():
    rng = np.random.default_rng(1)
    m1, m2 = rng.standard_normal(16), rng.standard_normal(16)
    v1, v2 = rng.uniform(0.1, 2.0, 16), rng.uniform(0.1, 2.0, 16)
    exp = ((m1 - m2) ** 2).sum() + ((np.sqrt(v1) - np.sqrt(v2)) ** 2).sum()
    assert abs(frechet_distance(m1, np.diag(v1), m2, np.diag(v2)) - exp) < 1e-9


def test_statistics_match_numpy_and_shapes_are_checked():
    rng = np.random.default_rng(2)
    a = rng.standard_normal((200, 12)).astype(np.float32)
    mu, sigma = activation_statistics(torch.tensor(a))
    assert np.allclose(mu, a.astype(np.float64).mean(0)) and np.allclose(sigma, np.cov(a.astype(np.float64), rowvar=False))
    with pytest.raises(ValueError):
        frechet_distance(np.zeros(3), np.eye(3), np.zeros(4), np.eye(4))
    with pytest.raises(ValueError):
        activation_statistics(torch.zeros(1, 4))
    b = a + 0.5
    d = fid_from_activations(torch.tensor(a), torch.tensor(b))
    assert abs(d - 12 * 0.25) < 1e-4      # same covariance, means shifted by 0.5 in 12 dimensions


def test_singular_product_takes_the_ridge_path_or_stays_finite():
    z = np.zeros((4, 4))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = frechet_distance(np.zeros(4), z, np.ones(4), z)
    assert np.isfinite(d) and abs(d - 4.0) < 1e-3


@pytest.mark.gpu
def test_statistics_on_the_device():
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(4096, 256, device="cuda", generator=g)
    mu, sigma = activation_statistics(a)
    an = a.double().cpu().numpy()
    assert np.allclose(mu, an.mean(0), atol=1e-12) and np.allclose(sigma, np.cov(an, rowvar=False), atol=1e-10)
    assert abs(fid_from_activations(a, a)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("N,D", [(1000, 2048), (777, 192), (33, 70)])
def test_in_tree_covariance_kernel_matches_numpy(N, D):
    """lg_fid_stats (csrc/fid.hip, fp64 matrix instruction) against np.mean / np.cov of the same fp32 activations, incl.
    a feature count that is not a multiple of the 64-wide tile and a sample count that is not a multiple of the slab."""
    g = torch.Generator(device="cuda").manual_seed(N)
    a = torch.randn(N, D, device="cuda", generator=g) * 3.0 + 1.5
    from littlegan_amd import ops
    mu, sigma = ops.fid_stats(a)
    an = a.double().cpu().numpy()
    assert np.abs(mu.cpu().numpy() - an.mean(0)).max() < 1e-12
    ref = np.cov(an, rowvar=False)
    assert np.abs(sigma.cpu().numpy() - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    assert torch.equal(sigma, sigma.t())


def test_evaluate_calc_on_saved_activations(tmp_path):
    """evaluate.py:29-59 on saved activations: pre-calculate writes {mu, sigma}; calc prints / returns the Frechet distance and
    appends one line in the reference's log format; identical activations give 0, a shifted set its closed form."""
    from littlegan_amd import fid
    rng = np.random.default_rng(5)
    real = rng.standard_normal((300, 16)).astype(np.float32)
    np.save(tmp_path / "real.npy", real)
    gen_dir = tmp_path / "gen"
    gen_dir.mkdir()
    shift = np.zeros(16, np.float32)
    shift[3] = 2.0
    np.save(gen_dir / "activations.npy", real + shift)          # same covariance, mean moved by 2 in one coordinate
    stats = str(tmp_path / "stats.npz")
    mu, sigma = fid.pre_calculate(str(tmp_path / "real.npy"), stats)
    assert np.abs(mu - real.astype(np.float64).mean(0)).max() < 1e-6 and sigma.shape == (16, 16)
    log = str(tmp_path / "fid.log")
    assert abs(fid.calc(str(tmp_path / "real.npy"), stats, log)) < 1e-6
    v = fid.calc(str(gen_dir), stats, log)
    assert abs(v - 4.0) < 1e-4
    lines = [ln for ln in open(log).read().split("\n") if ln.strip()]
    assert len(lines) == 2 and abs(float(lines[1].split()[-1]) - v) < 1e-9
