"""Diagnostic: per-phase s_memtime stamps of conv_down3 (build with LG_EXTRA_FLAGS=-DLG_D3_STAMPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
buf = torch.zeros(512 * 64, dtype=torch.int64, device="cuda")
os.environ["LG_D3_STAMPBUF"] = hex(buf.data_ptr())
from littlegan_amd import ops
B, dt = 256, 1
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
w = torch.randn(5, 5, 64, 128, device="cuda") * 0.05
pack = ops.conv_pack(w, 64, 128, dt)
x16 = torch.randn(B, 64, 64, 64, device="cuda").to(torch.bfloat16)
bias = torch.zeros(128, device="cuda")
for _ in range(3):
    ops.conv2d_s2_fwd_stats(None, pack, bias, 128, dt, gm, bt, x16=x16, z16=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv2d_s2_fwd_stats(None, pack, bias, 128, dt, gm, bt, x16=x16, z16=True)
e1.record()
torch.cuda.synchronize()
print(f"kernel time (events, incl. the stats-final launch): {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
st = buf.view(512, 64).cpu().numpy()
t0 = st[:, 0].min()
for b in (0, 1, 8, 9, 100, 256, 257, 511):
    row = st[b]
    n = int((row > 0).sum())
    d = [int(row[i] - row[i - 1]) for i in range(1, n)]
    print(f"block {b}: start +{int(row[0]-t0)} ; deltas {d}")
span = float((st.max(1) - st[:, 0]).max())
print("longest block span (ticks):", span, " all-block span:", float(st.max() - st[:, 0].min()))
print("first stamps (rel):", sorted((st[:, 0] - t0).tolist())[::64])
print("last stamps (rel):", sorted((st.max(1) - t0).tolist())[::64])
