"""Micro-benchmark (GPU) of single conv layers at the C3 shapes (B=256, 128x128), HIP-event timed.
   LG_DBG=<bits> enables the ablation switches of conv_halo.hip (results are then wrong, timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from littlegan_amd import ops
B = int(os.environ.get("LG_B", "256"))
dt = 1 if os.environ.get("LG_DT", "bf16") == "bf16" else 0
LAYERS = [  # name, kind, Hs, cb, cs
    ("convT1 fwd 8->16", "up", 8, 256, 384), ("convT2 fwd 16->32", "up", 16, 128, 256), ("convT3 fwd 32->64", "up", 32, 64, 128),
    ("convT4 fwd 64->128", "up", 64, 32, 64), ("conv2 fwd 64->32", "down", 32, 64, 128), ("conv3 fwd 32->16", "down", 16, 128, 256),
    ("conv4 fwd 16->8", "down", 8, 256, 384), ("final s1t 128", "s1t", 128, 3, 32), ("conv1 dgrad 64->128", "up", 64, 3, 64),
    ("wgrad 16/32 (128,256)", "wgrad", 16, 128, 256), ("wgrad 32/64 (64,128)", "wgrad", 32, 64, 128), ("wgrad 64/128 (32,64)", "wgrad", 64, 32, 64),
    ("wgrad 8/16 (256,384)", "wgrad", 8, 256, 384),
    ("convT1 dgrad 16->8", "ddown", 8, 256, 384), ("convT4 dgrad 128->64", "ddown", 64, 32, 64), ("convT3 dgrad 64->32", "ddown", 32, 64, 128), ("convT2 dgrad 32->16", "ddown", 16, 128, 256),
    ("conv2 dgrad 32->64", "dup", 32, 64, 128), ("conv3 dgrad 16->32", "dup", 16, 128, 256), ("conv4 dgrad 8->16", "dup", 8, 256, 384),
]
FUSE = os.environ.get("LG_FUSE", "0") == "1"  # data-gradient cases: with the norm-backward sums of the layer below in the epilogue
M16 = os.environ.get("LG_M16", "1") == "1" and dt == 1  # production bf16 path: sources read from their bf16 mirrors
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
sel = sys.argv[1:] or None
for name, kind, Hs, cb, cs in LAYERS:
    if sel and not any(s in name for s in sel):
        continue
    if dt == 0 and kind in ("ddown", "dup"):   # those rows time the bf16-mirror data gradients: no f32 form of that call
        continue
    w = torch.randn(5, 5, cb, cs, device="cuda") * 0.05
    pack = ops.conv_pack(w, cb, cs, dt)
    small = torch.randn(B, Hs, Hs, cs, device="cuda")
    big = torch.randn(B, 2 * Hs, 2 * Hs, cb, device="cuda") if kind != "s1t" else None
    bias_b, bias_s = torch.zeros(cb, device="cuda"), torch.zeros(cs, device="cuda")
    if kind == "up":
        out = torch.empty(B, 2 * Hs, 2 * Hs, cb, device="cuda")
        fn = (lambda: ops.convT_s2_fwd(small, pack, bias_b, cb, dt, out=out)) if cb != 3 else (lambda: ops.conv2d_s2_dgrad(small, pack, cb, dt, out=out))
        if M16 and cb != 3:
            s16 = small.to(torch.bfloat16)
            fn = lambda: ops.convT_s2_fwd_stats(None, pack, bias_b, cb, dt, gm, bt, x16=s16, z16=True)
    elif kind == "down":
        out = torch.empty(B, Hs, Hs, cs, device="cuda")
        fn = lambda: ops.conv2d_s2_fwd(big, pack, bias_s, cs, dt, out=out)
        if M16:
            b16 = big.to(torch.bfloat16)
            fn = lambda: ops.conv2d_s2_fwd_stats(None, pack, bias_s, cs, dt, gm, bt, x16=b16, z16=True)
    elif kind == "ddown":  # data gradient of a transposed conv (DOWN form): dz [B,2Hs,2Hs,cb] bf16 -> g [B,Hs,Hs,cs] bf16
        d16 = big.to(torch.bfloat16)
        fn = lambda: ops.convT_s2_dgrad(None, pack, cs, dt, dy16=d16, out_bf16=True)
        if FUSE:
            zf = small.to(torch.bfloat16)
            stf = ops.instnorm_stats(small, gm, bt, 0, 0.3)
            fn = lambda: ops.convT_s2_dgrad(None, pack, cs, dt, dy16=d16, out_bf16=True, fuse=(zf, stf, 0.3))
    elif kind == "dup":    # data gradient of a conv (UP form): dz [B,Hs,Hs,cs] bf16 -> g [B,2Hs,2Hs,cb] bf16
        d16 = small.to(torch.bfloat16)
        fn = lambda: ops.conv2d_s2_dgrad(None, pack, cb, dt, dy16=d16, out_bf16=True)
        if FUSE:
            zf = big.to(torch.bfloat16)
            stf = ops.instnorm_stats(big, gm, bt, 0, 0.3)
            fn = lambda: ops.conv2d_s2_dgrad(None, pack, cb, dt, dy16=d16, out_bf16=True, fuse=(zf, stf, 0.3))
    elif kind == "s1t":
        out = torch.empty(B, Hs, Hs, cb, device="cuda")
        fn = lambda: ops.convT_s1_tanh_fwd(small, pack, bias_b, cb, dt, out=out)
        if M16:
            s16 = small.to(torch.bfloat16)
            fn = lambda: ops.convT_s1_tanh_fwd(None, pack, bias_b, cb, dt, out=out, x16=s16)
            if os.environ.get("LG_Z16", "0") == "1":  # InstanceNorm + LeakyReLU applied while staging the raw conv output
                stz = ops.instnorm_stats(small, gm, bt, 0, 0.3)
                fn = lambda: ops.convT_s1_tanh_fwd_z16(s16, stz, 0.3, pack, bias_b, cb, dt, out=out)
    else:
        dw = torch.empty(5, 5, cb, cs, device="cuda")
        fn = lambda: ops.conv2d_s2_wgrad(big, small, dw, False, dt)
        if M16:
            b16, s16 = big.to(torch.bfloat16), small.to(torch.bfloat16)
            fn = lambda: ops.conv2d_s2_wgrad(None, None, dw, False, dt, x16=b16, dy16=s16)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 50.0 * B * Hs * Hs * cb * cs
    print(f"{name:26s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)
