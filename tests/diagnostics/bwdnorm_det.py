"""Diagnostic: is lg_convT_s2_dgrad_bn deterministic (a) with g = another tensor, (b) with g aliasing z (ONE buffer descriptor value)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops
ALPHA = 0.3
B, s, cb, cs = 32, 64, 32, 64
g_ = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *sh: torch.randn(*sh, generator=g_, device="cuda")
pack = ops.conv_pack(rnd(5, 5, cb, cs) * 0.05, cb, cs, 1)
shape = (B, 2 * s, 2 * s, cb)
z16 = (rnd(*shape) * 1.5 + rnd(B, 1, 1, 1)).to(torch.bfloat16)
g16 = rnd(*shape).to(torch.bfloat16)
st = ops.instnorm_stats(z16.float(), torch.tensor([0.9], device="cuda"), torch.tensor([0.15], device="cuda"), 0, ALPHA)
zl16 = (rnd(B, s, s, cs) * 1.3 + 0.2).to(torch.bfloat16)
stl = ops.instnorm_stats(zl16.float(), torch.tensor([1.1], device="cuda"), torch.tensor([-0.05], device="cuda"), 0, ALPHA)
coef = torch.zeros(B, 8, device="cuda")
coef[:, 0] = st[:, 0]; coef[:, 1] = st[:, 4]; coef[:, 2] = st[:, 2]; coef[:, 3] = st[:, 3]; coef[:, 4] = 0.01; coef[:, 5] = 0.02
for name, g in (("g = separate tensor", g16), ("g aliases z", z16)):
    outs = [ops.convT_s2_dgrad_bn(z16, g, coef, ALPHA, pack, cs, fuse=(zl16, stl, ALPHA))[0].clone() for _ in range(6)]
    torch.cuda.synchronize()
    print(name, ": launches equal to the first:", [bool(torch.equal(outs[0], o)) for o in outs[1:]],
          " differing elements:", [int((outs[0] != o).sum()) for o in outs[1:]])
