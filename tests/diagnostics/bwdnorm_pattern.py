"""Diagnostic (round 5): WHERE do launches of lg_convT_s2_dgrad_bn differ from bwd_apply16 + lg_convT_s2_dgrad_nf and from each other?
Run with LG_LIB_VARIANT naming a build of conv_down3.hip with -DLG_D3_COEF_PLAIN (the round-4 form that was not deterministic) and,
for comparison, with the product library.  Prints, per launch: differing elements, the items (tiles) they sit in with the item's
position k in its block's list and the block's index (upper half of the grid = the staggered blocks), histograms over the output
channel, the tile row / column, and whether the fused sums differ.  One process = one configuration; LG_REPS launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops

ALPHA = 0.3
B, s, cb, cs = int(os.environ.get("LG_B", "32")), int(os.environ.get("LG_S", "64")), 32, 64
REPS = int(os.environ.get("LG_REPS", "8"))
g_ = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *sh: torch.randn(*sh, generator=g_, device="cuda")
pack = ops.conv_pack(rnd(5, 5, cb, cs) * 0.05, cb, cs, 1)
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
pref = p_ref.buf.clone()
coef = ops.instnorm_bwd_coef(z16, st, P)
torch.cuda.synchronize()
print(f"variant={os.environ.get('LG_LIB_VARIANT')} blocks/CU={os.environ.get('LG_D3_BLOCKS_PER_CU', '2')} B={B} s={s}  ref kernel: {ops.last_kernel()}")
tpx, tpy = s // 16, s // 8
tpi = tpx * tpy
nitems = B * tpi
bpc = int(os.environ.get("LG_D3_BLOCKS_PER_CU", "2"))
G = min(nitems, bpc * 256)
outs = []
for rep in range(REPS):
    o, pp = ops.convT_s2_dgrad_bn(z16, g16, coef, ALPHA, pack, cs, fuse=(zl16, stl, ALPHA))
    outs.append((o.clone(), pp.buf.clone()))
torch.cuda.synchronize()
print("bn kernel:", ops.last_kernel(), " items", nitems, " grid", G)


def describe(o, tag):
    d = (o.float() - g_ref.float())
    nzm = d != 0
    n = int(nzm.sum())
    print(f"{tag}: differing elements {n} of {d.numel()}  max |diff| {float(d.abs().max()):.4g}  (ref max {float(g_ref.float().abs().max()):.3g})")
    if not n:
        return
    nz = nzm.nonzero()
    tiles = {}
    for n_, y_, x_, c_ in nz.tolist():
        key = (n_, y_ // 8, x_ // 16)
        t = tiles.setdefault(key, [0, set(), set(), set()])
        t[0] += 1; t[1].add(y_ % 8); t[2].add(x_ % 16); t[3].add(c_)
    print(f"  tiles touched: {len(tiles)}")
    for (n_, ty, tx), (cnt, rows, cols, chans) in sorted(tiles.items())[:24]:
        item = n_ * tpi + ty * tpx + tx
        lb, k = item % G, item // G
        q = G // 8
        bidx = (lb % q) * 8 + lb // q if G % 8 == 0 else -1
        full = cnt == 8 * 16 * cs
        print(f"    item {item:5d} (n {n_}, ty {ty}, tx {tx}) list pos k={k} lb={lb} blockIdx={bidx} upper-half={bidx >= (G + 1) // 2}"
              f"  elements {cnt}{' (WHOLE tile)' if full else ''} rows {sorted(rows)} cols {sorted(cols)} chans {len(chans)}: {sorted(chans)[:16]}")
    hist_c = torch.zeros(cs, dtype=torch.long); hist_c.index_add_(0, nz[:, 3].cpu(), torch.ones(len(nz), dtype=torch.long))
    print("  per-channel histogram:", hist_c.tolist())


for rep, (o, pb) in enumerate(outs):
    describe(o, f"launch {rep} vs apply+conv")
    print(f"   fused sums equal to the reference's: {bool(torch.equal(pb, pref))};  equal to launch 0: {bool(torch.equal(o, outs[0][0]))}")

