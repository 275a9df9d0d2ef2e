"""Timing of the FID activation statistics (fid.py:185-188) on synthetic Inception-sized activations [N, 2048]:
the in-tree fp64-MFMA kernel (lg_fid_stats) on the GPU, np.mean / np.cov on the host cores beside it, and the host
matrix square root of the Frechet distance (fid.py:144-163)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from littlegan_amd import ops
from littlegan_amd.fid import frechet_distance
for N in (10000, 50000):
    D = 2048
    a = torch.randn(N, D, device="cuda") * 0.5 + 0.3
    ops.fid_stats(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        mu, sg = ops.fid_stats(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    fl = 2.0 * N * D * (D + 64) / 2 * 1.0  # upper-triangular tiles only
    an = a.cpu().numpy().astype(np.float64)
    t0 = time.perf_counter(); ref = np.cov(an, rowvar=False); tc = time.perf_counter() - t0
    err = np.abs(sg.cpu().numpy() - ref).max()
    print(f"[N={N}, D={D}] lg_fid_stats {dt*1e3:.2f} ms ({fl/dt/1e12:.1f} TFLOP/s fp64, triangular) | numpy cov on {os.cpu_count()} host threads {tc*1e3:.0f} ms | max |sigma - np.cov| {err:.1e}", flush=True)
m1, s1 = mu.cpu().numpy(), sg.cpu().numpy()
t0 = time.perf_counter(); d = frechet_distance(m1, s1, m1 + 0.01, s1 * 1.01); ts = time.perf_counter() - t0
print(f"host sqrtm + trace (scipy) for D=2048: {ts:.1f} s, d^2 = {d:.4f}")
