"""Round-5 layout experiment (verdict item 3): conv_down3 fed with a CHANNEL-SLICE-MAJOR source [B][C/16][H][W][16] instead of
NHWC.  The kernel stages its halo in 16-channel slices: from NHWC a slice uses 32 of a pixel's 2 C bytes per pass (a wave's 64 pieces
touch 32 cache lines and use a quarter of each at C = 64), from the slice-major tensor its 64 pieces are 2 KB contiguous.
Run twice: LG_D3_SLICE_MAJOR unset (writes the reference outputs) and =1 (permutes the inputs with torch, compares bit for bit, times).
usage: python scripts/probe/d3_slice_major.py <dir>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops
out_dir = sys.argv[1]
SM = bool(os.environ.get("LG_D3_SLICE_MAJOR"))
B = int(os.environ.get("LG_B", "256"))
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
CASES = [("conv2 fwd 64x64x64 -> 32x32x128", "fwd", 32, 64, 128, B), ("conv3 fwd 32x32x128 -> 16x16x256", "fwd", 16, 128, 256, B),
         ("conv2 fwd at 2B", "fwd", 32, 64, 128, 2 * B), ("convT3 dgrad 64x64x64 -> 32x32x128", "dgrad", 32, 64, 128, B),
         ("convT4 dgrad 128x128x32 -> 64x64x64", "dgrad", 64, 32, 64, B), ("convT4 dgrad at 2B", "dgrad", 64, 32, 64, 2 * B)]


def to_sm(x):   # [B, H, W, C] -> the same storage size, laid out [B][C/16][H][W][16]
    b, h, w, c = x.shape
    return x.view(b, h, w, c // 16, 16).permute(0, 3, 1, 2, 4).contiguous().view(b, h, w, c)


for ci, (name, kind, Hs, cb, cs, b_) in enumerate(CASES):
    g = torch.Generator(device="cuda").manual_seed(100 + ci)
    w = torch.randn(5, 5, cb, cs, device="cuda", generator=g) * 0.05
    pack = ops.conv_pack(w, cb, cs, 1)
    big16 = torch.randn(b_, 2 * Hs, 2 * Hs, cb, device="cuda", generator=g).to(torch.bfloat16)
    src = to_sm(big16) if SM else big16
    bias = torch.zeros(cs, device="cuda")
    if kind == "fwd":
        fn = lambda: ops.conv2d_s2_fwd_stats(None, pack, bias, cs, 1, gm, bt, x16=src, z16=True, defer_stats=True)[0]
    else:
        fn = lambda: ops.convT_s2_dgrad(None, pack, cs, 1, dy16=src, out_bf16=True)
    o = fn()
    kern = ops.last_kernel()
    torch.cuda.synchronize()
    ref_path = os.path.join(out_dir, f"ref_{ci}.pt")
    if SM:
        ref = torch.load(ref_path).cuda()
        same = bool(torch.equal(ref, o[:8]))
    else:
        torch.save(o[:8].cpu(), ref_path)
        same = None
    for _ in range(3):
        fn()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    ts.sort()
    print(f"{'slice-major' if SM else 'NHWC       '} {name:42s} {kern:28s} median {ts[2]:7.1f} us  min {ts[0]:7.1f}" + ("" if same is None else f"  bit-equal to NHWC: {same}"), flush=True)
