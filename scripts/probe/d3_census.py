"""Diagnostic: where does conv_down3's time go that the in-kernel tap stamps do not show?
Build with LG_EXTRA_FLAGS=-DLG_D3_STAMPS (scripts/probe/d3_census.sh).  Per block: HW_ID / XCC_ID (which CU), the chip-wide
100 MHz clock at its first and last instruction, and wave 0's s_memtime stamps (start | per slice: taps done, barrier passed |
per item: epilogue done).  Prints: blocks per CU, start / end spread against the kernel's HIP-event duration, in-block
phase sums, the clock (s_memtime ticks per 10-ns real tick)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

buf = torch.zeros(512 * 64, dtype=torch.int64, device="cuda")
os.environ["LG_D3_STAMPBUF"] = hex(buf.data_ptr())
from littlegan_amd import ops  # noqa: E402

B, dt = int(os.environ.get("LG_B", "256")), 1
which = sys.argv[1] if len(sys.argv) > 1 else "conv2"
cb, cs, Hs = {"conv2": (64, 128, 32), "conv3": (128, 256, 16)}[which]
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
w = torch.randn(5, 5, cb, cs, device="cuda") * 0.05
pack = ops.conv_pack(w, cb, cs, dt)
x16 = torch.randn(B, 2 * Hs, 2 * Hs, cb, device="cuda").to(torch.bfloat16)
bias = torch.zeros(cs, device="cuda")
run = lambda: ops.conv2d_s2_fwd_stats(None, pack, bias, cs, dt, gm, bt, x16=x16, z16=True)
import time  # noqa: E402
t_end = time.time() + 2.5   # >= 2 s of back-to-back launches: the clock the chip HOLDS under this kernel, not its ramp from idle
while time.time() < t_end:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print(f"{which} forward B={B}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call by HIP events (conv + the stats-final launch)")
st = buf.view(512, 64).cpu().numpy().astype(np.int64)   # stamps of the LAST call
hw, xcc = st[:, 61] & 0xffffffff, st[:, 61] >> 32
cu, sh, se, simd = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 4) & 3
per_cu = collections.Counter(zip((xcc & 15).tolist(), se.tolist(), sh.tolist(), cu.tolist()))
nb = int((st[:, 62] > 0).sum())
st, hw, xcc, cu, sh, se = st[:nb], hw[:nb], xcc[:nb], cu[:nb], sh[:nb], se[:nb]
per_cu = collections.Counter(zip((xcc & 15).tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print("blocks:", len(st), " distinct CUs:", len(per_cu), " blocks per CU histogram:", dict(collections.Counter(per_cu.values())))
r0, r1 = st[:, 62], st[:, 63]
t0 = r0.min()
print(f"start of blocks after the first block (us): median {np.median(r0 - t0) / 100:.2f}  p90 {np.percentile(r0 - t0, 90) / 100:.2f}  max {(r0 - t0).max() / 100:.2f}")
print(f"end   of blocks (us): min {(r1 - t0).min() / 100:.2f}  median {np.median(r1 - t0) / 100:.2f}  max {(r1 - t0).max() / 100:.2f}   block life median {np.median(r1 - r0) / 100:.2f} us")
half = len(st) // 2
if len(st) == 512:
    print(f"first block of a CU (blockIdx < 256) ends at median {np.median(r1[:half] - t0) / 100:.2f} us, the second (>= 256) at {np.median(r1[half:] - t0) / 100:.2f} us")
    order = np.argsort(r1)
    print("blockIdx >= 256 among the 128 earliest finishers:", int((order[:128] >= 256).sum()), " among the 128 latest:", int((order[-128:] >= 256).sum()))
    # same-CU pairs: how far apart do the two blocks of one CU finish?
    key = [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc & 15, se, sh, cu)]
    ends = collections.defaultdict(list)
    for k_, e_ in zip(key, (r1 - t0)):
        ends[k_].append(e_)
    gaps = np.array([abs(v[0] - v[1]) for v in ends.values() if len(v) == 2]) / 100
    print(f"|end(block A) - end(block B)| on one CU: median {np.median(gaps):.2f} us  p90 {np.percentile(gaps, 90):.2f} us")
    xe = collections.defaultdict(list)
    for x_, e_ in zip((xcc & 15).tolist(), (r1 - t0)):
        xe[x_].append(e_)
    print("last block end per XCD (us):", {k_: round(max(v) / 100, 1) for k_, v in sorted(xe.items())})
late = (r0 - t0) > 0.25 * (r1 - t0).max()
print(f"blocks that start later than 25 % into the kernel: {int(late.sum())}")
# in-block phases of wave 0 (s_memtime ticks = shader cycles)
clk_b = (st[:, 60] - st[:, 59]) / np.maximum(1, r1 - r0) * 0.1   # GHz per block: shader cycles per 10-ns tick
print(f"clock held under the kernel (block first..last, s_memtime / s_memrealtime): median {np.median(clk_b):.3f} GHz  min {clk_b.min():.3f}  max {clk_b.max():.3f}")
mf_ = 100 * (cb // 16) * (B * Hs * Hs // 128) * (cs // 128) / len(st)
life_ = np.median(st[:, 60] - st[:, 59])
print(f"block life {life_:.0f} cycles; MFMAs per wave per block {mf_:.0f} -> matrix pipe busy over the block life with 2 waves per SIMD: {2 * mf_ * 32 / life_ * 100:.0f} %")
if os.environ.get("LG_D3_STAMPS_LITE"):
    sys.exit(0)
nst = (st[:, :59] > 0).sum(1)
ticks_per_real = []
tap, bar, epi = [], [], []
for b in range(len(st)):
    row = st[b, :nst[b]]
    if len(row) < 3:
        continue
    ticks_per_real.append((row[-1] - row[0]) / max(1, (r1[b] - r0[b])))
    d = np.diff(row)
    # sequence after the start stamp: (taps, barrier)[, epilogue] per slice
    i, items = 0, 0
    while i + 1 < len(d):
        tap.append(d[i]); bar.append(d[i + 1]); i += 2
        # an epilogue stamp follows the last slice of an item: detect by the slice count (Cs / 16 slices per item)
        items += 1
        if items % (cb // 16) == 0 and i < len(d):
            epi.append(d[i]); i += 1
clk = np.median(ticks_per_real) * 100e6 / 1e9
print(f"clock by s_memtime / s_memrealtime: {clk:.2f} GHz")
tap, bar, epi = np.array(tap), np.array(bar), np.array(epi)
per_block = np.median(nst)
print(f"stamps per block {per_block:.0f}; per slice: taps {np.median(tap):.0f} cycles (ideal alone 3200, two waves sharing a SIMD 6400), "
      f"commit + barrier {np.median(bar):.0f}; per item: epilogue {np.median(epi):.0f}")
life = np.median(st[:, :59].max(1) - st[:, 0])
print(f"block life {life:.0f} cycles = taps {tap.sum() / len(st):.0f} + barriers {bar.sum() / len(st):.0f} + epilogues {epi.sum() / len(st):.0f} (sums per block)")
mf = 100 * (cb // 16) * (B * Hs * Hs // 128) * (cs // 128) / len(st)   # MFMAs per wave per block
print(f"MFMAs per wave per block {mf:.0f} -> pipe busy if 2 waves per SIMD over the block life: {2 * mf * 32 / life * 100:.0f} %, if 1: {mf * 32 / life * 100:.0f} %")
