"""A/B per layer (GPU): stride-2 forward conv fed with the normalised bf16 map (+ the apply pass that wrote it) against the
normalising form fed with the raw map (lg_conv2d_s2_fwd_stats_zn), at the shapes of D on the Adjuster's output (2B = 512)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops
B = int(os.environ.get("LG_B", "512"))
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, H, cb, cs in (("conv2", 64, 64, 128), ("conv3", 32, 128, 256), ("conv4", 16, 256, 384)):
    w = torch.randn(5, 5, cb, cs, device="cuda") * 0.05
    pack = ops.conv_pack(w, cb, cs, 1)
    bias = torch.zeros(cs, device="cuda")
    z16 = (torch.randn(B, H, H, cb, device="cuda") * 1.3 + 0.2).to(torch.bfloat16)
    st = ops.instnorm_stats(z16.float(), gm, bt, 0, 0.3)
    h16 = torch.empty_like(z16)
    t_apply = timed(lambda: ops.instnorm_apply(z16, st, None, 0, 1, 0.3, out16=h16, want_f32=False))
    t_conv = timed(lambda: ops.conv2d_s2_fwd_stats(None, pack, bias, cs, 1, gm, bt, x16=h16, z16=True))
    if not ops.conv2d_s2_fwd_stats_zn_supported(B, H, H, cb, cs, 1):   # (the sample-pair tiling of the 8 x 8 level has no normalising form)
        print(f"{name}: apply {t_apply:7.1f} us + conv {t_conv:7.1f} us = {t_apply + t_conv:7.1f}   no normalising form")
        continue
    t_zn = timed(lambda: ops.conv2d_s2_fwd_stats_zn(z16, st, 0.3, pack, bias, cs, 1, gm, bt))
    print(f"{name}: apply {t_apply:7.1f} us + conv {t_conv:7.1f} us = {t_apply + t_conv:7.1f}   normalising conv {t_zn:7.1f} us   net {t_zn - t_apply - t_conv:+7.1f}")
