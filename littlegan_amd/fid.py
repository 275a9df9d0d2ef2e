"""Fréchet-distance arithmetic of the FID pass (/root/reference/fid.py:112-163 and :185-188).

Only the arithmetic: d^2 = |mu1 - mu2|^2 + tr(S1 + S2 - 2 (S1 S2)^(1/2)).  The Inception pool_3 activations it is
normally fed with need a frozen graph that the reference downloads (fid.py:276) and this pipeline cannot obtain, so
the entry points take activation matrices.  The O(N D^2) part (mean and covariance of [N, D] activations) runs on the
device in fp64 (in-tree kernel on the fp64 matrix instruction, csrc/fid.hip); the D x D matrix square root stays on the host (scipy), as in the reference."""
import warnings

import numpy as np
import torch


def activation_statistics(act: torch.Tensor):
    """fid.py:185-188: mu = mean over samples, sigma = np.cov(act, rowvar=False) (divisor N - 1).  act [N, D].
    On the GPU this is the in-tree fp64-MFMA kernel (csrc/fid.hip, lg_fid_stats); a host tensor (tests, tiny inputs) is
    reduced with torch in float64."""
    if act.dim() != 2 or act.shape[0] < 2:
        raise ValueError("activation_statistics: need an [N >= 2, D] matrix")
    if act.is_cuda:
        from . import ops
        mu, sigma = ops.fid_stats(act.to(torch.float32).contiguous())
        return mu.cpu().numpy(), sigma.cpu().numpy()
    a = act.to(torch.float64)
    mu = a.mean(dim=0)
    c = a - mu
    sigma = (c.t() @ c) / (a.shape[0] - 1)
    return mu.cpu().numpy(), sigma.cpu().numpy()


def frechet_distance(mu1, sigma1, mu2, sigma2, eps: float = 1e-6) -> float:
    """fid.py:112-163.  Same guards as the reference: a non-finite square root of the product retries with eps on both
    diagonals (with a warning); a complex result is accepted only if its diagonal is real to 1e-3."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(np.asarray(mu1, np.float64)), np.atleast_1d(np.asarray(mu2, np.float64))
    s1, s2 = np.atleast_2d(np.asarray(sigma1, np.float64)), np.atleast_2d(np.asarray(sigma2, np.float64))
    if mu1.shape != mu2.shape:
        raise ValueError("frechet_distance: mean vectors have different lengths")
    if s1.shape != s2.shape:
        raise ValueError("frechet_distance: covariances have different dimensions")
    root = linalg.sqrtm(s1 @ s2)
    if isinstance(root, tuple):  # old scipy returns (sqrtm, error estimate) when disp=False; be liberal
        root = root[0]
    if not np.isfinite(root).all():
        warnings.warn(f"fid calculation produces singular product; adding {eps} to diagonal of cov estimates")
        ridge = np.eye(s1.shape[0]) * eps
        root = linalg.sqrtm((s1 + ridge) @ (s2 + ridge))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0.0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(root.imag))}")
        root = root.real
    d = mu1 - mu2
    return float(d @ d + np.trace(s1) + np.trace(s2) - 2.0 * np.trace(root))


def fid_from_activations(act_a: torch.Tensor, act_b: torch.Tensor) -> float:
    """FID between two activation sets (what evaluate.py computes once the Inception features exist)."""
    m1, s1 = activation_statistics(act_a)
    m2, s2 = activation_statistics(act_b)
    return frechet_distance(m1, s1, m2, s2)


# ------------------------------------------------------------------ evaluate.py:29-59 on saved activations
def load_activations(path: str) -> torch.Tensor:
    """[N, D] float32 Inception pool_3 activations from `path`: an .npy / .npz file (key "act", else its first array), or a
    directory holding activations.npy.  The reference computes them from the JPEGs of `image_path` with a frozen Inception graph
    it downloads (evaluate.py:45-46,53-55, fid.py:36-106,276); that graph cannot be obtained in this pipeline, so the entry points
    below start from the activations a user has saved."""
    import os
    if os.path.isdir(path):
        path = os.path.join(path, "activations.npy")
    a = np.load(path)
    if hasattr(a, "files"):
        a = a["act"] if "act" in a.files else a[a.files[0]]
    a = np.asarray(a, np.float32)
    if a.ndim != 2 or a.shape[0] < 2:
        raise ValueError(f"{path}: need an [N >= 2, D] activation matrix, got {a.shape}")
    t = torch.from_numpy(a)
    return t.cuda() if torch.cuda.is_available() else t


def pre_calculate(act_path: str, stats_path: str):
    """evaluate.py `pre-calculate` (:29-42): statistics of the real images' activations -> stats npz {mu, sigma}."""
    mu, sigma = activation_statistics(load_activations(act_path))
    np.savez_compressed(stats_path, mu=mu, sigma=sigma)
    print("finished")
    return mu, sigma


def calc(act_path: str, stats_path: str, output_file: str) -> float:
    """evaluate.py `calc` (:43-59): statistics of the generated images' activations (on the device: lg_fid_stats), the Frechet
    distance to the pre-calculated statistics, "FID: <value>" on stdout and one line appended to the log in the reference's
    format ("\\n <iso time> <value>\\n ")."""
    import datetime
    with np.load(stats_path) as f:
        mu_real, sigma_real = f["mu"][:], f["sigma"][:]
    mu_gen, sigma_gen = activation_statistics(load_activations(act_path))
    fid_value = frechet_distance(mu_gen, sigma_gen, mu_real, sigma_real)
    print("FID: %s" % fid_value)
    with open(output_file, "a") as f:
        print("\n", datetime.datetime.now().isoformat(), fid_value, end="\n ", file=f)
    return fid_value
