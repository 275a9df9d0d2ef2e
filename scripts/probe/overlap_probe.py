"""Probe: do the HBM-bound norm passes overlap with the latency-bound conv kernels when issued on two streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops

dt = 1
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
gm, bt = torch.ones(1, device=dev), torch.zeros(1, device=dev)
# conv2 fwd: [B,64,64,64] -> [B,32,32,128] ; convT3 fwd: [B,32,32,128] -> [B,64,64,64]
w2 = rn(5, 5, 64, 128) * 0.05; p2 = ops.conv_pack(w2, 64, 128, dt); b2 = torch.zeros(128, device=dev)
x2 = rn(B, 64, 64, 64).to(torch.bfloat16)
w3 = rn(5, 5, 64, 128) * 0.05; p3 = ops.conv_pack(w3, 64, 128, dt); b3 = torch.zeros(64, device=dev)
x3 = rn(B, 32, 32, 128).to(torch.bfloat16)
# norm pass on a big map: z [B,128,128,32] bf16
z = rn(B, 128, 128, 32).to(torch.bfloat16)
st = ops.instnorm_stats(z.float(), gm, bt, 0, 0.3)
h16 = torch.empty_like(z)
gg = rn(B, 128, 128, 32).to(torch.bfloat16)
d16 = torch.empty_like(z)
dgm, dbt, db = torch.empty(1, device=dev), torch.empty(1, device=dev), torch.empty(32, device=dev)

def conv_work():
    ops.conv2d_s2_fwd_stats(None, p2, b2, 128, dt, gm, bt, x16=x2, z16=True)
    ops.convT_s2_fwd_stats(None, p3, b3, 64, dt, gm, bt, x16=x3, z16=True)

def norm_work():
    ops.instnorm_apply(z, st, None, 0, 1, 0.3, out16=h16, want_f32=False)
    ops.instnorm_bwd(z, st, gg, dgm, dbt, 0, 1, 0.3, out16=d16, want_f32=False, db=db)

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1):
        conv_work()
    with torch.cuda.stream(s2):
        norm_work()
tc, tn = timeit(conv_work), timeit(norm_work)
tb = timeit(both)
print(f"B={B}: conv {tc:.3f} ms, norm {tn:.3f} ms, sum {tc + tn:.3f}, concurrent on two streams {tb:.3f} ms")
