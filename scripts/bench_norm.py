"""Micro-benchmark (GPU) of the InstanceNorm passes of the bf16 activation path at the C3 map sizes (B=256): achieved HBM GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from littlegan_amd import ops
B = int(os.environ.get("LG_B", "256"))
MAPS = [(128, 32), (64, 64), (32, 128), (16, 256)]
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")


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


for H, C in MAPS:
    z = torch.randn(B, H, H, C, device="cuda")
    z16 = z.to(torch.bfloat16)
    g16 = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
    sk16 = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
    st = ops.instnorm_stats(z, gm, bt, 0, 0.3)
    h16, dz16 = torch.empty_like(z16), torch.empty_like(z16)
    dg, dbeta, db = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(C, device="cuda")
    nb = z16.numel() * 2
    cases = [
        ("apply   z16 -> h16", lambda: ops.instnorm_apply(z16, st, None, 0, 1, 0.3, out16=h16, want_f32=False), 2 * nb),
        ("apply+skip16 -> h16", lambda: ops.instnorm_apply(z16, st, sk16, 0, 1, 0.3, out16=h16, want_f32=False), 3 * nb),
        ("bwd  (partial+apply)", lambda: ops.instnorm_bwd(z16, st, g16, dg, dbeta, 0, 1, 0.3, out16=dz16, want_f32=False), 5 * nb),
        ("bwd+db", lambda: ops.instnorm_bwd(z16, st, g16, dg, dbeta, 0, 1, 0.3, out16=dz16, want_f32=False, db=db), 5 * nb),
        ("torch bf16 copy", lambda: h16.copy_(z16), 2 * nb),
    ]
    for name, fn, bytes_ in cases:
        us = timeit(fn)
        print(f"{H:4d}x{H:<4d}x{C:<4d} {name:24s} {us:8.1f} us  {bytes_ / us / 1e3:8.1f} GB/s", flush=True)
