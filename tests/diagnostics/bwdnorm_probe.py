"""Diagnostic (round 5): read the operand the non-deterministic BWDNORM build really staged WITHOUT touching the kernel.
With one-hot weights the data gradient copies its operand: W[1][1][ci][ci] = 1, W[2][2][ci][32 + ci] = 1 gives
out[y][x][ci] = dz[2y][2x][ci], out[y][x][32 + ci] = dz[2y + 1][2x + 1][ci] (one exact product per output, fp32 accumulation of
zeros, bf16 store of a bf16 value) — and a second weight set the other two pixel parities.  So the output of the UNMODIFIED variant
library (LG_LIB_VARIANT=d3plain: conv_down3.hip with -DLG_D3_COEF_PLAIN) is the dz image each tile held in LDS; it is compared with
bwd_apply16's dz16, and for every wrong 8-channel piece the script searches which altered coefficient record reproduces the staged bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops

ALPHA = 0.3
B, s, cb, cs = int(os.environ.get("LG_B", "32")), 64, 32, 64
REPS = int(os.environ.get("LG_REPS", "6"))
g_ = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *sh: torch.randn(*sh, generator=g_, device="cuda")
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
coef = ops.instnorm_bwd_coef(z16, st, P)
torch.cuda.synchronize()
print(f"variant={os.environ.get('LG_LIB_VARIANT')} B={B}")
eye = torch.eye(cb, device="cuda")
names = ["mu", "mul", "a", "b", "m1", "m2", "m1l", "m2l"]


def bf(x):
    return (x.to(torch.int32) << 16).view(torch.float32)


def dz_of(z8, g8, rec):   # lg_bwdnorm8, every operation separately rounded in fp32
    m_, ml, a, b, m1, m2, m1l, m2l = [rec[..., i:i + 1] for i in range(8)]
    c = (z8 - m_) - ml
    p = torch.where(a * c + b > 0, g8, ALPHA * g8)
    d = a * ((((p - m1) - m1l) - c * m2) - c * m2l)
    return d.to(torch.bfloat16).view(torch.int16)


zi, gi, di = z16.view(torch.int16), g16.view(torch.int16), dz16.view(torch.int16)
for wset, taps in enumerate((((1, 1), (2, 2)), ((1, 2), (2, 1)))):
    w = torch.zeros(5, 5, cb, cs, device="cuda")
    for j, (ky, kx) in enumerate(taps):
        w[ky, kx, :, 32 * j:32 * j + 32] = eye
    pack = ops.conv_pack(w, cb, cs, 1)
    # expected: out[y][x][32 j + ci] = dz[2y + ky - 1][2x + kx - 1][ci]
    exp = torch.cat([di[:, (ky - 1)::2, (kx - 1)::2, :] for (ky, kx) in taps], -1)
    for rep in range(REPS):
        o, _ = ops.convT_s2_dgrad_bn(z16, g16, coef, ALPHA, pack, cs, fuse=(zl16, stl, ALPHA))
        torch.cuda.synchronize()
        oi = o.view(torch.int16)
        bad = (oi != exp)
        pieces = bad.view(B, s, s, 8, 8).any(-1)   # 8-channel pieces = the kernel's 16-byte staging units
        print(f"weights {wset} launch {rep} ({ops.last_kernel()}): wrong elements {int(bad.sum())}, wrong 16-byte pieces {int(pieces.sum())}")
        runs = {}
        for n_, y_, x_, p_ in pieces.nonzero().tolist()[:400]:
            j, half = p_ // 4, p_ % 4           # j: which tap / parity; half: 8-channel group of the 32 source channels
            ky, kx = taps[j]
            sy, sx = 2 * y_ + ky - 1, 2 * x_ + kx - 1
            runs.setdefault((n_, sy, half), []).append(sx)
        for (n_, sy, half), sxs in sorted(runs.items())[:20]:
            print(f"   sample {n_} source row {sy} channels {8 * half}..{8 * half + 7}: source columns {sorted(sxs)}")
        shown = 0
        for n_, y_, x_, p_ in pieces.nonzero().tolist():
            if shown >= 10: break
            shown += 1
            j, half = p_ // 4, p_ % 4
            ky, kx = taps[j]
            sy, sx = 2 * y_ + ky - 1, 2 * x_ + kx - 1
            gotv = oi[n_, y_, x_, 8 * p_:8 * p_ + 8]
            expv = di[n_, sy, sx, 8 * half:8 * half + 8]
            z8, g8 = bf(zi[n_, sy, sx, 8 * half:8 * half + 8]), bf(gi[n_, sy, sx, 8 * half:8 * half + 8])
            base = coef[n_]
            ok = torch.equal(dz_of(z8, g8, base), expv)
            print(f"   piece sample {n_} source ({sy}, {sx}) ch {8 * half}..: staged {[f'{v:.5g}' for v in bf(gotv).tolist()]}")
            print(f"        dz16 {[f'{v:.5g}' for v in bf(expv).tolist()]}   (torch restatement == dz16: {ok})")
            found = []
            cands = torch.cat([coef, torch.zeros(1, 8, device="cuda")], 0)
            for f in range(8):
                rec = base.repeat(B + 1, 1); rec[:, f] = cands[:, f]
                hit = [h for h in (dz_of(z8[None], g8[None], rec) == gotv[None]).all(-1).nonzero().flatten().tolist() if h != n_]
                if hit: found.append((names[f] + " of sample", hit[:8]))
            for lo, hi, nm in ((0, 4, "first 16 bytes (mu, mul, a, b) of sample"), (4, 8, "second 16 bytes (m1, m2, m1l, m2l) of sample"), (0, 8, "whole record of sample")):
                rec = base.repeat(B + 1, 1); rec[:, lo:hi] = cands[:, lo:hi]
                hit = [h for h in (dz_of(z8[None], g8[None], rec) == gotv[None]).all(-1).nonzero().flatten().tolist() if h != n_]
                if hit: found.append((nm, hit[:8]))
            print("        reproduced by:", found if found else "nothing in the search space")
            # per-element forensics: which element, the implied additive error in front of the final scale a, and the record's small fields
            e = int((gotv != expv).nonzero().flatten()[0])
            dd = float(bf(gotv)[e] - bf(expv)[e]) / float(base[2])
            m_, ml, a, b, m1, m2, m1l, m2l = [base[i] for i in range(8)]
            c = (z8[e] - m_) - ml
            pe = g8[e] if float(a * c + b) > 0 else ALPHA * g8[e]
            alts = {"m2 in place of m1l": a * ((((pe - m1) - m2) - c * m2) - c * m2l), "m1 not subtracted": a * (((pe - m1l) - c * m2) - c * m2l),
                    "m1 subtracted twice": a * (((((pe - m1) - m1) - m1l) - c * m2) - c * m2l), "b in place of m1": a * ((((pe - b) - m1l) - c * m2) - c * m2l),
                    "mul in place of m1l": a * ((((pe - m1) - ml) - c * m2) - c * m2l), "c m2 term dropped": a * (((pe - m1) - m1l) - c * m2l),
                    "c m2 term doubled": a * (((((pe - m1) - m1l) - c * m2) - c * m2) - c * m2l), "m2l in place of m1l": a * ((((pe - m1) - m2l) - c * m2) - c * m2l),
                    "c without mul": a * ((((pe - m1) - m1l) - (z8[e] - m_) * m2) - (z8[e] - m_) * m2l), "m1 of the element's neighbour formula (p uses alpha branch flipped)": a * (((((ALPHA * g8[e] if float(a * c + b) > 0 else g8[e]) - m1) - m1l) - c * m2) - c * m2l)}
            hits = [k_ for k_, v_ in alts.items() if int(v_.to(torch.bfloat16).view(torch.int16)) == int(gotv[e])]
            print(f"        element {e}: (staged - right) / a = {dd:.3e}   m1 {float(m1):.3e} m2 {float(m2):.3e} m1l {float(m1l):.2e} m2l {float(m2l):.2e} mul {float(ml):.2e} c {float(c):.3f} c*m2 {float(c * m2):.3e}  -> {hits}")
