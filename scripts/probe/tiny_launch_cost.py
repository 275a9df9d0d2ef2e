"""GPU probe: what does a tiny (4.8-us) launch cost BETWEEN two persistent conv launches?  A chain of conv2-forward launches (B = 256,
~105 us each) with k = 0 .. 4 `lg_adam_advance` launches (one wave, a few scalar operations) behind every conv; HIP-event time per conv
of the chain, eager and as a captured graph.  Round 3 found removing 22 such launches from the step time-neutral; this measures the
marginal cost directly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from littlegan_amd import ops

B = 256
w = torch.randn(5, 5, 64, 128, device="cuda") * 0.05
pack = ops.conv_pack(w, 64, 128, 1)
x16 = torch.randn(B, 64, 64, 64, device="cuda").to(torch.bfloat16)
bias = torch.zeros(128, device="cuda")
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
state = torch.ones(4, device="cuda")
NCONV = 40


def chain(k):
    for _ in range(NCONV):
        ops.conv2d_s2_fwd_stats(None, pack, bias, 128, 1, gm, bt, x16=x16, z16=True, defer_stats=True)
        for _ in range(k):
            ops.adam_advance(state, 0.9, 0.999)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / NCONV)
    ts.sort()
    return ts[len(ts) // 2]


base = None
for k in range(5):
    t_e = timed(lambda: chain(k))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(k); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            chain(k)
    t_g = timed(g.replay)
    if base is None:
        base = (t_e, t_g)
    print(f"k={k}: eager {t_e:7.1f} us per conv (+{(t_e - base[0]) / max(k, 1):5.2f} per tiny launch)   graph {t_g:7.1f} us (+{(t_g - base[1]) / max(k, 1):5.2f})", flush=True)
