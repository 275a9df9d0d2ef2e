"""Diagnostic: where does lg_convT_s2_dgrad_bn differ from bwd_apply16 + lg_convT_s2_dgrad_nf?  (small shape, prints the pattern)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops

ALPHA = 0.3
B, s, cb, cs = int(os.environ.get("LG_B", "4")), int(os.environ.get("LG_S", "16")), 32, 64
g_ = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *sh: torch.randn(*sh, generator=g_, device="cuda")
w = rnd(5, 5, cb, cs) * 0.05
pack = ops.conv_pack(w, cb, cs, 1)
shape = (B, 2 * s, 2 * s, cb)
z16 = (rnd(*shape) * 1.5 + rnd(B, 1, 1, 1)).to(torch.bfloat16)
g16 = rnd(*shape).to(torch.bfloat16)
gm, bt = torch.tensor([0.9], device="cuda"), torch.tensor([0.15], device="cuda")
st = ops.instnorm_stats(z16.float(), gm, bt, 0, ALPHA)
zl16 = (rnd(B, s, s, cs) * 1.3 + 0.2).to(torch.bfloat16)
stl = ops.instnorm_stats(zl16.float(), torch.tensor([1.1], device="cuda"), torch.tensor([-0.05], device="cuda"), 0, ALPHA)
zz, gg = z16.double().reshape(B, 4, -1), g16.double().reshape(B, 4, -1)
mu = (st[:, 0].double() + st[:, 4].double()).view(B, 1, 1)
c32 = (z16.float().reshape(B, 4, -1) - st[:, 0].view(B, 1, 1)) - st[:, 4].view(B, 1, 1)
gp = torch.where(st[:, 2].view(B, 1, 1) * c32 + st[:, 3].view(B, 1, 1) > 0, gg, ALPHA * gg)
sums = torch.stack([gp.sum(-1), (gp * (zz - mu)).sum(-1)], -1).contiguous()
P = ops.NormPartials(sums.view(torch.uint8).reshape(-1), 4, ALPHA, shape)
dz16 = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
ops.instnorm_bwd(z16, st, g16, None, None, 0, 1, ALPHA, out16=dz16, want_f32=False, partials=P)
g_ref, p_ref = ops.convT_s2_dgrad(None, pack, cs, 1, dy16=dz16, out_bf16=True, fuse=(zl16, stl, ALPHA))
print("ref kernel:", ops.last_kernel())
coef = ops.instnorm_bwd_coef(z16, st, P)
g_bn, p_bn = ops.convT_s2_dgrad_bn(z16, g16, coef, ALPHA, pack, cs, fuse=(zl16, stl, ALPHA))
print("bn kernel:", ops.last_kernel())
torch.cuda.synchronize()
d = (g_bn.float() - g_ref.float()).abs()
print("equal:", torch.equal(g_bn, g_ref), " max diff", float(d.max()), " differing elements", int((d > 0).sum()), "of", d.numel(), " ref max", float(g_ref.float().abs().max()))
nz = (d > 0).nonzero()
print("first differing (n, y, x, c):", nz[:12].tolist())
print("per-sample count:", [(int((d[n] > 0).sum())) for n in range(B)])
print("per-row count sample 0:", [(int((d[0, y] > 0).sum())) for y in range(s)])
print("per-col count sample 0:", [(int((d[0, :, x] > 0).sum())) for x in range(s)])
# the dz the BN kernel must have formed: recompute with torch in fp32 in the kernel's order and compare with bwd_apply16's dz16
co = coef
sh = (B, 1, 1, 1)
zf, gf = z16.float(), g16.float()
c = (zf - co[:, 0].view(sh)) - co[:, 1].view(sh)
a_, b_ = co[:, 2].view(sh), co[:, 3].view(sh)
gp32 = torch.where(a_ * c + b_ > 0, gf, ALPHA * gf)
dd = a_ * ((((gp32 - co[:, 4].view(sh)) - co[:, 6].view(sh)) - c * co[:, 5].view(sh)) - c * co[:, 7].view(sh))
print("torch fp32 dz vs bwd_apply16 dz16: differing", int((dd.to(torch.bfloat16) != dz16).sum()), "of", dz16.numel())
g_bn2, _ = ops.convT_s2_dgrad_bn(z16, g16, coef, ALPHA, pack, cs, fuse=(zl16, stl, ALPHA))
print("BN deterministic across two launches:", torch.equal(g_bn, g_bn2))
g_ref2, _ = ops.convT_s2_dgrad(None, pack, cs, 1, dy16=dz16, out_bf16=True, fuse=(zl16, stl, ALPHA))
print("reference deterministic:", torch.equal(g_ref, g_ref2))
tiles = {}
for n_, y_, x_, c_ in nz.tolist():
    tiles[(n_, y_ // 8, x_ // 16)] = tiles.get((n_, y_ // 8, x_ // 16), 0) + 1
print("affected tiles (n, ty, tx): count", sorted(tiles.items())[:40])
G = 512
tpi = (s // 8) * (s // 16)
print("item index of affected tiles and its position in its block's list:", sorted({((n_ * tpi + ty * (s // 16) + tx), (n_ * tpi + ty * (s // 16) + tx) // G) for (n_, ty, tx) in tiles})[:40])
