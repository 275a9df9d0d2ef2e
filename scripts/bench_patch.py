"""Micro-benchmark (GPU) of the 3-channel-source layers (conv1 forward with fused moments; final-layer data gradient) at B=256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from littlegan_amd import ops
B = int(os.environ.get("LG_B", "256"))
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


img = torch.rand(B, 128, 128, 3, device="cuda") * 2 - 1
w1 = torch.randn(5, 5, 3, 64, device="cuda") * 0.05
p1 = ops.conv_pack(w1, 3, 64, 1)
b1 = torch.zeros(64, device="cuda")
print(f"conv1 fwd (z16 + moments)   {timeit(lambda: ops.conv2d_s2_fwd_stats(img, p1, b1, 64, 1, gm, bt, z16=True)):8.1f} us", flush=True)
wf = torch.randn(5, 5, 3, 32, device="cuda") * 0.05
pf = ops.conv_pack(wf, 3, 32, 1)
dpre = torch.randn(B, 128, 128, 3, device="cuda")
dx16 = torch.empty(B, 128, 128, 32, dtype=torch.bfloat16, device="cuda")
print(f"final dgrad (bf16 out)      {timeit(lambda: ops.convT_s1_tanh_bwd(None, dpre, pf, 32, 1, dx16=dx16)):8.1f} us", flush=True)
z = torch.randn(B, 128, 128, 32, device="cuda")
z16 = z.to(torch.bfloat16)
st = ops.instnorm_stats(z, gm, bt, 0, 0.3)
print(f"final dgrad + norm sums     {timeit(lambda: ops.convT_s1_tanh_bwd(None, dpre, pf, 32, 1, dx16=dx16, fuse=(z16, st, 0.3))):8.1f} us", flush=True)
dz16 = torch.randn(B, 64, 64, 64, device="cuda").to(torch.bfloat16)
print(f"conv1 dgrad (image grad)    {timeit(lambda: ops.conv2d_s2_dgrad(None, p1, 3, 1, dy16=dz16)):8.1f} us", flush=True)
dw1 = torch.empty(5, 5, 3, 64, device="cuda")
print(f"conv1 wgrad                 {timeit(lambda: ops.conv2d_s2_wgrad(img, None, dw1, False, 1, dy16=dz16)):8.1f} us", flush=True)
h16 = torch.randn(B, 128, 128, 32, device="cuda").to(torch.bfloat16)
dwf, dbf = torch.empty(5, 5, 3, 32, device="cuda"), torch.empty(3, device="cuda")
print(f"final wgrad                 {timeit(lambda: ops.convT_s1_tanh_bwd(None, dpre, pf, 32, 1, dw=dwf, db=dbf, x16=h16)):8.1f} us", flush=True)
bf = torch.zeros(3, device="cuda")
yo = torch.empty(B, 128, 128, 3, device="cuda")
print(f"final fwd (h16 in)          {timeit(lambda: ops.convT_s1_tanh_fwd(None, pf, bf, 3, 1, out=yo, x16=h16)):8.1f} us", flush=True)
print(f"final fwd (raw z16 + norm)  {timeit(lambda: ops.convT_s1_tanh_fwd_z16(z16, st, 0.3, pf, bf, 3, 1, out=yo)):8.1f} us", flush=True)
