"""Probe: does running the two norm-backward passes on sample chunks (data still in the 256 MiB Infinity Cache between
the partial-sums pass and the apply pass) beat one pass over the whole batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops
dev = "cuda"
B = 256
for (H, C) in ((128, 32), (64, 64), (32, 128)):
    z = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    g = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    gm, bt = torch.ones(1, device=dev), torch.zeros(1, device=dev)
    st = ops.instnorm_stats(z.float(), gm, bt, 0, 0.3)
    d16 = torch.empty_like(z)
    h16 = torch.empty_like(z)
    dgm, dbt, db = torch.empty(1, device=dev), torch.empty(1, device=dev), torch.empty(C, device=dev)
    def bwd(nchunk):
        cs = B // nchunk
        for k in range(nchunk):
            sl = slice(k * cs, (k + 1) * cs)
            ops.instnorm_bwd(z[sl], st[sl], g[sl], dgm, dbt, 0, 1, 0.3, out16=d16[sl], want_f32=False, db=db, accumulate=(k > 0))
    def timeit(fn, n=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    print(f"map {H}x{H}x{C} B={B}: bwd whole {timeit(lambda: bwd(1)):.0f} us, 2 chunks {timeit(lambda: bwd(2)):.0f}, 4 chunks {timeit(lambda: bwd(4)):.0f}, 8 chunks {timeit(lambda: bwd(8)):.0f}")
