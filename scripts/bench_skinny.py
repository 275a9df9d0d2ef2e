"""Micro-benchmark (GPU) of the dense / discriminator-head GEMMs at the C3 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from littlegan_amd import ops


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


K, c = 24576, 40
for B in (256, 512):
    x = torch.randn(B, K, device="cuda")
    wpr, wc = torch.randn(K, 1, device="cuda") * 0.02, torch.randn(K, c, device="cuda") * 0.02
    bpr, bc = torch.zeros(1, device="cuda"), torch.zeros(c, device="cuda")
    dz = torch.randn(B, 1 + c, device="cuda")
    dwpr, dbpr, dwc, dbc = torch.empty_like(wpr), torch.empty_like(bpr), torch.empty_like(wc), torch.empty_like(bc)
    print(f"B={B} heads_fwd   {timeit(lambda: ops.heads_fwd(x, wpr, bpr, wc, bc)):8.1f} us")
    print(f"B={B} heads_dgrad {timeit(lambda: ops.heads_dgrad(dz, wpr, wc)):8.1f} us")
    print(f"B={B} heads_wgrad {timeit(lambda: ops.heads_wgrad(x, dz, dwpr, dbpr, dwc, dbc)):8.1f} us")
B, Kd = 256, 133
xd, w, b = torch.randn(B, Kd, device="cuda"), torch.randn(Kd, K, device="cuda") * 0.1, torch.zeros(K, device="cuda")
dy, dw, db = torch.randn(B, K, device="cuda"), torch.empty(Kd, K, device="cuda"), torch.empty(K, device="cuda")
print(f"B={B} dense_fwd   {timeit(lambda: ops.dense_fwd(xd, w, b)):8.1f} us")
print(f"B={B} dense_wgrad {timeit(lambda: ops.dense_wgrad(xd, dy, dw, db)):8.1f} us")
