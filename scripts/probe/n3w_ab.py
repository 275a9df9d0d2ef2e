"""A/B (GPU): weight gradient of the final 3-channel layer (n3_wgrad16, 32 channels, stride 1) and of Encoder.conv1 (64 channels, stride 2)
at the launch shapes; LG_N3W_TH8=1 = 8-row tiles everywhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops


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


B = 256
x16 = torch.randn(B, 128, 128, 32, device="cuda").to(torch.bfloat16)
dpre = torch.randn(B, 128, 128, 3, device="cuda")
w = torch.randn(5, 5, 3, 32, device="cuda") * 0.05
pack = ops.conv_pack(w, 3, 32, 1)
dw, db = torch.zeros(5, 5, 3, 32, device="cuda"), torch.zeros(3, device="cuda")
t = timed(lambda: ops.convT_s1_tanh_bwd(None, dpre, pack, 32, 1, dw=dw, db=db, x16=x16))
print(f"final layer dw+db, B={B}: {t:.1f} us  ({ops.last_kernel()})")
img = torch.randn(2 * B, 128, 128, 3, device="cuda")
dz16 = torch.randn(2 * B, 64, 64, 64, device="cuda").to(torch.bfloat16)
dw1 = torch.zeros(5, 5, 3, 64, device="cuda")
t = timed(lambda: ops.conv2d_s2_wgrad(img, None, dw1, False, 1, dy16=dz16))
print(f"conv1 dw, 2B={2 * B}: {t:.1f} us  ({ops.last_kernel()})")
