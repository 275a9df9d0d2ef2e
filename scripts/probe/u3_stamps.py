"""Diagnostic: per-phase s_memtime stamps of conv_up3 (build with LG_EXTRA_FLAGS=-DLG_U3_STAMPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
buf = torch.zeros(256 * 8 * 32, dtype=torch.int64, device="cuda")
os.environ["LG_U3_STAMPBUF"] = hex(buf.data_ptr())
from littlegan_amd import ops
B, dt = 256, 1
which = sys.argv[1] if len(sys.argv) > 1 else "t3"
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
if which == "t3":
    cb, cs, Hs = 64, 128, 32
else:
    cb, cs, Hs = 32, 64, 64
w = torch.randn(5, 5, cb, cs, device="cuda") * 0.05
pack = ops.conv_pack(w, cb, cs, dt)
x16 = torch.randn(B, Hs, Hs, cs, device="cuda").to(torch.bfloat16)
bias = torch.zeros(cb, device="cuda")
import time
t_end = time.time() + 2.0   # the clock the chip holds under the kernel, not its ramp from idle
while time.time() < t_end:
    for _ in range(50):
        ops.convT_s2_fwd_stats(None, pack, bias, cb, dt, gm, bt, x16=x16, z16=True)
    torch.cuda.synchronize()
NW = 4 if (which != "t3" and not os.environ.get("LG_U3_T4_8W")) else 8
NB = 256 * 8 // NW
st = buf.view(NB, NW, 32).cpu().numpy()
names = ["start", "class", "staged", "bar1", "rows", "(bar2+next start)"]
for b in (0, 100):
    for w in range(NW):
        row = st[b, w]
        n = int((row > 0).sum())
        d = [int(row[i] - row[i - 1]) for i in range(1, min(n, 16))]
        print(f"block {b} wave {w}: {d}")
print("stamp order per item:", names)
import numpy as np
# per wave role: median cycles of each phase over all blocks and items (5 stamps per item)
for w in range(NW):
    ph = [[] for _ in range(5)]
    for b in range(NB):
        row = st[b, w]; n = int((row > 0).sum())
        d = np.diff(row[:n])
        for i in range(len(d)):
            ph[i % 5].append(d[i])
    print(f"wave {w}: median cycles  compute {np.median(ph[0]):.0f}  stage {np.median(ph[1]):.0f}  wait-barrier1 {np.median(ph[2]):.0f}  commit+rows {np.median(ph[3]):.0f}  barrier2+issue {np.median(ph[4]):.0f}   item total {sum(np.median(x) for x in ph):.0f}")
print("block 0 wave 0 total cycles:", int(st[0, 0][st[0, 0] > 0].max() - st[0, 0, 0]))
