"""GPU diagnostic: conv2 data gradient (conv_up3<128,64>) with and without the fused norm sums — the gradient must be bit-identical;
prints where it is not (sample, output row, output column, channel octet)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops
B = int(os.environ.get("LG_B", "4"))
torch.manual_seed(0)
w = torch.randn(5, 5, 64, 128, device="cuda") * 0.05
pack = ops.conv_pack(w, 64, 128, 1)
dy16 = torch.randn(B, 32, 32, 128, device="cuda").to(torch.bfloat16)
z16 = (torch.randn(B, 64, 64, 64, device="cuda") * 1.7 + 0.4).to(torch.bfloat16)
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
st = ops.instnorm_stats(z16.float(), gm, bt, 0, 0.3)
g0 = ops.conv2d_s2_dgrad(None, pack, 64, 1, dy16=dy16, out_bf16=True)
print("plain :", ops.last_kernel())
g1, parts = ops.conv2d_s2_dgrad(None, pack, 64, 1, dy16=dy16, out_bf16=True, fuse=(z16, st, 0.3))
print("fused :", ops.last_kernel(), None if parts is None else parts.nparts)
torch.cuda.synchronize()
d = (g0.float() != g1.float())
print("mismatching elements:", int(d.sum()), "of", d.numel())
if d.any():
    idx = d.nonzero()
    print("samples:", sorted(set(idx[:, 0].tolist()))[:16])
    print("rows   :", sorted(set(idx[:, 1].tolist()))[:64])
    print("cols   :", sorted(set(idx[:, 2].tolist()))[:64])
    print("chan/8 :", sorted(set((idx[:, 3] // 8).tolist())))
    n, y, x, c = idx[0].tolist()
    print("first:", (n, y, x, c), float(g0[n, y, x, c]), float(g1[n, y, x, c]))
    # is the fused value some OTHER element of the plain result?  (misplaced piece)
    v = g1[n, y, x, (c // 8) * 8:(c // 8) * 8 + 8]
    m = (g0.view(-1, 8) == v).all(1).nonzero()
    print("piece found at flat piece index:", m[:4].flatten().tolist(), "expected", ((n * 64 + y) * 64 + x) * 8 + c // 8)
if parts is not None:
    import ctypes
    buf = parts.buf[: B * parts.nparts * 16].view(torch.float64).view(B, parts.nparts, 2)
    # reference sums from the plain gradient
    s32 = st.float()
    zf = z16.float()
    cc = (zf - s32[:, 0].view(-1, 1, 1, 1)) - s32[:, 4].view(-1, 1, 1, 1)
    t = s32[:, 2].view(-1, 1, 1, 1) * cc + s32[:, 3].view(-1, 1, 1, 1)
    gp = torch.where(t > 0, g0.float(), 0.3 * g0.float())
    S1 = gp.double().sum((1, 2, 3)); S2 = (gp * cc).double().sum((1, 2, 3))
    print("S1 fused / ref:", buf[:, :, 0].sum(1).tolist()[:4], S1.tolist()[:4])
    print("S2 fused / ref:", buf[:, :, 1].sum(1).tolist()[:4], S2.tolist()[:4])
if d.any():
    # per tile (8 x 16 source pixels -> 16 x 32 output pixels): mismatch counts; items are dealt lb + step * G
    tiles = d.view(B, 4, 16, 2, 32, 64).sum((2, 4, 5)).view(-1)
    bad = tiles.nonzero().flatten().tolist()
    print("tiles with mismatches:", len(bad), "of", tiles.numel(), "first:", bad[:24], "last:", bad[-8:])
    print("all mismatching fused values zero:", bool((g1[d] == 0).all()), " plain values there zero:", int((g0[d] == 0).sum()))
    t0 = bad[0]; n, ty, tx = t0 // 8, (t0 % 8) // 2, t0 % 2
    sub = d[n, ty * 16:(ty + 1) * 16, tx * 32:(tx + 1) * 32]
    print("rows of first bad tile:", sub.any(2).any(1).nonzero().flatten().tolist(), "octets:", sub.view(16, 32, 8, 8).any(3).any(1).any(0).nonzero().flatten().tolist())
