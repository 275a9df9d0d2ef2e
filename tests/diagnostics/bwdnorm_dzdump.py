"""Diagnostic (round 5): WHICH staged operand pieces of lg_convT_s2_dgrad_bn are wrong in the non-deterministic build, and what were they
computed from?  Needs a variant library built with -DLG_D3_COEF_PLAIN -DLG_D3_DZDUMP (conv_down3.hip): every 16-byte piece of dz the
kernel writes to LDS is also written to a dump buffer [item][slice][piece].  The dump is compared with bwd_apply16's dz16 (the two are
bit-identical when the kernel is right); for every wrong piece the script searches which ALTERED coefficient record reproduces the
bits that were staged: one field / one 16-byte half / the whole record taken from another sample (or zero)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from littlegan_amd import ops

ALPHA = 0.3
B, s, cb, cs = int(os.environ.get("LG_B", "32")), 64, 32, 64
REPS = int(os.environ.get("LG_REPS", "6"))
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
g_ref, _ = ops.convT_s2_dgrad(None, pack, cs, 1, dy16=dz16, out_bf16=True, fuse=(zl16, stl, ALPHA))
coef = ops.instnorm_bwd_coef(z16, st, P)
torch.cuda.synchronize()

tpx, tpy = s // 16, s // 8
tpi = tpx * tpy
nitems = B * tpi
NQ, PPT = 19 * 70, 6
dump = torch.zeros(nitems * 2 * PPT * 256 * 8, dtype=torch.int16, device="cuda")
os.environ["LG_D3_DBGBUF"] = hex(dump.data_ptr())

# expected pieces [item, slice, q, 8]
item = torch.arange(nitems, device="cuda")
n_i, tt = item // tpi, item % tpi
y0, x0 = (tt // tpx) * 8, (tt % tpx) * 16
q = torch.arange(NQ, device="cuda")
hy, rem = q // 70, q % 70
hx, half = rem >> 1, rem & 1
sy = (2 * y0 - 1)[:, None] + hy[None, :]
sx = (2 * x0 - 1)[:, None] + hx[None, :]


def padded5(t):   # [B, H, W, 32] bf16 -> [B, H + 3, W + 3, 4, 8] int16 with the TF-SAME zero border (1 before, 2 after)
    return F.pad(t.view(torch.int16), (0, 0, 1, 2, 1, 2)).view(B, 2 * s + 3, 2 * s + 3, 4, 8)


def gather(t5):
    grp = torch.stack([half, 2 + half], 0)   # [slice, q]
    return t5[n_i[:, None, None], (sy + 1)[:, None, :], (sx + 1)[:, None, :], grp[None, :, :]]


exp = gather(padded5(dz16))
zp, gp5 = gather(padded5(z16)), gather(padded5(g16))
inside = ((sy >= 0) & (sy < 2 * s) & (sx >= 0) & (sx < 2 * s))[:, None, :].expand(-1, 2, -1)


def bf(x):   # int16 bit patterns -> float32 values
    return (x.to(torch.int32) << 16).view(torch.float32)


def dz_of(z8, g8, rec):   # lg_bwdnorm8 in torch fp32, every operation separately rounded; rec: [..., 8]
    m_, ml, a, b, m1, m2, m1l, m2l = [rec[..., i:i + 1] for i in range(8)]
    c = (z8 - m_) - ml
    p = torch.where(a * c + b > 0, g8, ALPHA * g8)
    d = a * ((((p - m1) - m1l) - c * m2) - c * m2l)
    return d.to(torch.bfloat16).view(torch.int16)


for rep in range(REPS):
    dump.zero_()
    o, _ = ops.convT_s2_dgrad_bn(z16, g16, coef, ALPHA, pack, cs, fuse=(zl16, stl, ALPHA))
    torch.cuda.synchronize()
    nd = int((o != g_ref).sum())
    got = dump.view(nitems, 2, PPT * 256, 8)[:, :, :NQ]
    bad = (got != exp).any(-1)
    print(f"launch {rep}: output elements differing from apply + conv: {nd};  staged pieces differing from dz16: {int(bad.sum())} of {bad.numel()}")
    if rep == 0:   # sanity of the harness: where the kernel is right the dump must equal dz16
        print("   (pieces equal:", int((~bad).sum()), ")")
    for it_, c_, q_ in bad.nonzero().tolist()[:24]:
        n_ = it_ // tpi
        tid, u = q_ % 256, q_ // 256
        z8, g8 = bf(zp[it_, c_, q_]), bf(gp5[it_, c_, q_])
        gotv, expv = got[it_, c_, q_], exp[it_, c_, q_]
        print(f"  item {it_} (sample {n_}) slice {c_} piece q={q_}: thread {tid} (wave {tid >> 6} lane {tid & 63}) u={u} halo ({int(hy[q_])}, {int(hx[q_])}) half {int(half[q_])} inside {bool(inside[it_, c_, q_])}")
        print(f"     staged {[f'{v:.5g}' for v in bf(gotv).tolist()]}")
        print(f"     dz16   {[f'{v:.5g}' for v in bf(expv).tolist()]}")
        base = coef[n_]
        assert torch.equal(dz_of(z8, g8, base) if inside[it_, c_, q_] else torch.zeros(8, dtype=torch.int16, device='cuda'), expv), "harness: torch restatement != dz16"
        found = []
        cands = torch.cat([coef, torch.zeros(1, 8, device="cuda")], 0)   # every sample's record + an all-zero one
        names = ["mu", "mul", "a", "b", "m1", "m2", "m1l", "m2l"]
        for f in range(8):
            rec = base.repeat(B + 1, 1); rec[:, f] = cands[:, f]
            hit = (dz_of(z8[None], g8[None], rec) == gotv[None]).all(-1).nonzero().flatten().tolist()
            hit = [h for h in hit if h != n_]
            if hit: found.append((names[f], hit[:6]))
        for lo, hi, nm in ((0, 4, "first 16 bytes (mu, mul, a, b)"), (4, 8, "second 16 bytes (m1, m2, m1l, m2l)"), (0, 8, "whole record")):
            rec = base.repeat(B + 1, 1); rec[:, lo:hi] = cands[:, lo:hi]
            hit = [h for h in (dz_of(z8[None], g8[None], rec) == gotv[None]).all(-1).nonzero().flatten().tolist() if h != n_]
            if hit: found.append((nm, hit[:6]))
        # other pieces' data with the right record: the neighbouring piece / the other slice / g and z swapped
        alt = {"z and g swapped": dz_of(g8, z8, base)}
        if q_ + 1 < NQ: alt["data of piece q + 1"] = dz_of(bf(zp[it_, c_, q_ + 1]), bf(gp5[it_, c_, q_ + 1]), base)
        if q_ >= 1: alt["data of piece q - 1"] = dz_of(bf(zp[it_, c_, q_ - 1]), bf(gp5[it_, c_, q_ - 1]), base)
        alt["data of the other slice"] = dz_of(bf(zp[it_, 1 - c_, q_]), bf(gp5[it_, 1 - c_, q_]), base)
        for k_, v_ in alt.items():
            if torch.equal(v_, gotv): found.append((k_, []))
        print("     reproduced by:", found if found else "NOTHING in the search space (one field / one half / whole record from another sample or zero; neighbouring data)")
