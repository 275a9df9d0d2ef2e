"""Diagnostic: does the side-stream clock probe (lg_clock_probe, bench.py's `clock` field) see what a busy kernel sees?
3 s of back-to-back conv2-forward launches (conv_down3, MFMA-dense) with the probe wave resident beside them; prints the
probe's 1-ms windows in 250-ms buckets next to the HIP-event time per conv call in the same buckets, then in-stream samples
(lg_clock_sample) taken behind every 10th call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from littlegan_amd import _lib as lib
from littlegan_amd import ops

L = lib.load()
B, dt = 256, 1
cb, cs, Hs = 64, 128, 32
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
w = torch.randn(5, 5, cb, cs, device="cuda") * 0.05
pack = ops.conv_pack(w, cb, cs, dt)
x16 = torch.randn(B, 2 * Hs, 2 * Hs, cb, device="cuda").to(torch.bfloat16)
bias = torch.zeros(cs, device="cuda")
run = lambda: ops.conv2d_s2_fwd_stats(None, pack, bias, cs, dt, gm, bt, x16=x16, z16=True)
for _ in range(5):
    run()
torch.cuda.synchronize()
time.sleep(1.0)   # idle first: the probe's first windows show the idle clock

out5 = torch.zeros(5, dtype=torch.int64, device="cuda")
series = torch.zeros(8192, dtype=torch.int32, device="cuda")
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    lib.check(L.lg_clock_probe(out5.data_ptr(), flag.data_ptr(), 6000, series.data_ptr(), series.numel(), side.cuda_stream), "probe")
time.sleep(0.2)
t0 = time.time()
evs = []
out3 = torch.zeros(3, dtype=torch.int64, device="cuda")
n = 0
while time.time() - t0 < 3.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    lib.check(L.lg_clock_sample(out3.data_ptr(), 20, torch.cuda.current_stream().cuda_stream), "sample")
    evs.append((time.time() - t0, e0, e1))
    n += 1
    if n % 20 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
t_load = time.time() - t0
lib.check(L.lg_clock_stop(flag.data_ptr(), torch.cuda.current_stream().cuda_stream), "stop")
torch.cuda.synchronize()
o = out5.tolist()
s = series[:min(o[4], series.numel())].cpu().numpy() / 1e3
print(f"probe: {o[4]} windows, mean {o[0] / max(o[1], 1) * 100:.1f} MHz; load ran {t_load:.2f} s after ~0.2 s of idle windows")
for k in range(0, len(s), 250):
    seg = s[k:k + 250]
    print(f"  windows {k:5d}..{k + len(seg):5d}: mean {seg.mean():7.1f}  min {seg.min():7.1f}  max {seg.max():7.1f} MHz")
us = np.array([e0.elapsed_time(e1) * 100 for _, e0, e1 in evs])   # us per call (10 calls per pair)
ts = np.array([t for t, _, _ in evs])
for lo in np.arange(0, 3.0, 0.25):
    m = (ts >= lo) & (ts < lo + 0.25)
    if m.any():
        print(f"  host time {lo:4.2f}..{lo + 0.25:4.2f} s: conv2 forward {us[m].mean():6.1f} us per call (HIP events, {m.sum()} groups)")
o3 = out3.tolist()
print(f"in-stream samples: {o3[2]} samples, mean {o3[0] / max(o3[1], 1) * 100:.1f} MHz")
