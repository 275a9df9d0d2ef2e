"""Probe (round 5): a level's weight gradient and its data gradient both read dz and nothing of each other — issued on two streams,
does one fill the other's tail?  Serial (one stream) against two streams, per level of the C3 step (B = 256, bf16 mirrors)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops

B = int(os.environ.get("LG_B", "256"))
ALPHA = 0.3
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
# (name, kind, cb, cs, small side)  conv: x[B,2s,2s,cb] -> z[B,s,s,cs]
LEVELS = [("enc.conv2", "conv", 64, 128, 32), ("enc.conv3", "conv", 128, 256, 16), ("enc.conv4", "conv", 256, 384, 8),
          ("dec.conv1", "convT", 256, 384, 8), ("dec.conv2", "convT", 128, 256, 16), ("dec.conv3", "convT", 64, 128, 32), ("dec.conv4", "convT", 32, 64, 64)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


side = torch.cuda.Stream()
tot_s = tot_p = 0.0
for name, kind, cb, cs, s in LEVELS:
    g = torch.Generator(device="cuda").manual_seed(1)
    w = torch.randn(5, 5, cb, cs, generator=g, device="cuda") * 0.05
    pack = ops.conv_pack(w, cb, cs, 1)
    big16 = torch.randn(B, 2 * s, 2 * s, cb, generator=g, device="cuda").to(torch.bfloat16)
    small16 = torch.randn(B, s, s, cs, generator=g, device="cuda").to(torch.bfloat16)
    dw = torch.empty(5, 5, cb, cs, device="cuda")
    zl16 = big16 if kind == "conv" else small16
    stl = ops.instnorm_stats(zl16.float(), gm, bt, 0, ALPHA)

    def dgrad():
        if kind == "conv":
            ops.conv2d_s2_dgrad(None, pack, cb, 1, dy16=small16, out_bf16=True, fuse=(zl16, stl, ALPHA))
        else:
            ops.convT_s2_dgrad(None, pack, cs, 1, dy16=big16, out_bf16=True, fuse=(zl16, stl, ALPHA))

    def wgrad():
        if kind == "conv":
            ops.conv2d_s2_wgrad(None, None, dw, False, 1, x16=big16, dy16=small16)
        else:
            ops.convT_s2_wgrad(None, None, dw, False, 1, x16=small16, dy16=big16)

    def serial():
        wgrad(); dgrad()

    def two():
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            wgrad()
            done = torch.cuda.Event(); done.record()
        dgrad()
        torch.cuda.current_stream().wait_event(done)

    td, tw, ts, tp = timeit(dgrad), timeit(wgrad), timeit(serial), timeit(two)
    tot_s += ts; tot_p += tp
    print(f"{name:10s} dgrad {td:7.1f}  wgrad {tw:7.1f}  serial {ts:7.1f}  two streams {tp:7.1f} us  ({100 * (tp / ts - 1):+.1f} %)", flush=True)
print(f"all levels: serial {tot_s:.1f} us, two streams {tot_p:.1f} us ({100 * (tot_p / tot_s - 1):+.1f} %)")
