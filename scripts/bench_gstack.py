"""The north-star number: forward of the Generator's transposed-conv stack (model.py:37-51,86-87; SURVEY.md a2 + a3) at B = 256, bf16,
128x128 — the five conv kernels exactly as the step launches them (bf16 source mirror -> bf16 z + fused InstanceNorm moments for the
four stride-2 levels; the row-sliding tanh layer for the final one), HIP-event timed per launch, rounds interleaved in ONE process.
Algorithmic work 50*B*Hs*Ws*cb*cs FLOP per layer = 423 GFLOP for the stack; peak 2.5 PFLOP/s dense bf16.

usage: python scripts/bench_gstack.py [out.json]   (LG_B=256, LG_ROUNDS=30)"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from littlegan_amd import _lib, ops  # noqa: E402

B = int(os.environ.get("LG_B", "256"))
ROUNDS = int(os.environ.get("LG_ROUNDS", "30"))
PEAK = 2500.0
LAYERS = [("convT1 8->16", 8, 256, 384), ("convT2 16->32", 16, 128, 256), ("convT3 32->64", 32, 64, 128), ("convT4 64->128", 64, 32, 64)]
gm, bt = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
g = torch.Generator(device="cuda").manual_seed(3)
fns, flops, names = [], [], []
for name, Hs, cb, cs in LAYERS:
    w = torch.randn(5, 5, cb, cs, device="cuda", generator=g) * 0.05
    pack = ops.conv_pack(w, cb, cs, 1)
    x16 = torch.randn(B, Hs, Hs, cs, device="cuda", generator=g).to(torch.bfloat16)   # random data (guide rule 25)
    bias = torch.randn(cb, device="cuda", generator=g) * 0.1
    # defer_stats=True: as the step launches them — the moment partials are finished inside the following apply launch, not by a launch of their own
    fns.append(lambda pack=pack, x16=x16, bias=bias, cb=cb: ops.convT_s2_fwd_stats(None, pack, bias, cb, 1, gm, bt, x16=x16, z16=True, defer_stats=True))
    flops.append(50.0 * B * Hs * Hs * cb * cs)
    names.append(name)
w = torch.randn(5, 5, 3, 32, device="cuda", generator=g) * 0.05
packf = ops.conv_pack(w, 3, 32, 1)
xf = torch.randn(B, 128, 128, 32, device="cuda", generator=g).to(torch.bfloat16)
bf = torch.zeros(3, device="cuda")
outf = torch.empty(B, 128, 128, 3, device="cuda")
fns.append(lambda: ops.convT_s1_tanh_fwd(None, packf, bf, 3, 1, out=outf, x16=xf))
flops.append(50.0 * B * 128 * 128 * 3 * 32)
names.append("final s1 tanh 128")

# CHAINED mode (LG_CHAINED=1, the default): the stack as Generator.__call__ runs it (littlegan_amd/model.py) — every conv reads the bf16
# map the InstanceNorm + LeakyReLU apply launch has just written from the previous conv's z (the apply launches run between the
# timed convs, untimed), so each layer meets its input in the cache state of the real step.  LG_CHAINED=0: every layer on a
# static input tensor of its own (the round-2 / round-3 method: all inputs cold, ~800 MB cycle through the caches per round).
CHAINED = os.environ.get("LG_CHAINED", "1") != "0"
if CHAINED:
    state = {"x": None}
    x0 = torch.randn(B, 8, 8, 384, device="cuda", generator=g).to(torch.bfloat16)
    packs, biases = [], []
    g2 = torch.Generator(device="cuda").manual_seed(3)
    for name, Hs, cb, cs in LAYERS:
        w = torch.randn(5, 5, cb, cs, device="cuda", generator=g2) * 0.05
        packs.append(ops.conv_pack(w, cb, cs, 1))
        biases.append(torch.randn(cb, device="cuda", generator=g2) * 0.1)
    h16 = [torch.empty(B, 2 * Hs, 2 * Hs, cb, dtype=torch.bfloat16, device="cuda") for _, Hs, cb, cs in LAYERS]
    zst = [None] * 4

    def conv_i(i):
        cb = LAYERS[i][2]
        src = x0 if i == 0 else h16[i - 1]
        zst[i] = ops.convT_s2_fwd_stats(None, packs[i], biases[i], cb, 1, gm, bt, x16=src, z16=True, defer_stats=True)

    def apply_i(i):   # untimed: InstanceNorm + LeakyReLU, moments finished inside this launch (as in the step)
        z, st = zst[i]
        ops.instnorm_apply(z, st, None, 0, 1, 0.3, out16=h16[i], want_f32=False)

    fns = [lambda i=i: conv_i(i) for i in range(4)] + [lambda: ops.convT_s1_tanh_fwd(None, packf, bf, 3, 1, out=outf, x16=h16[3])]
    between = [lambda i=i: apply_i(i) for i in range(4)] + [lambda: None]
else:
    between = [lambda: None] * len(fns)

kernels = []
for f, bt_ in zip(fns, between):   # warm-up + the kernel template each layer runs on
    for _ in range(3):
        f()
    kernels.append(_lib.load().lg_last_kernel().decode())
    bt_()
torch.cuda.synchronize()
times = [[] for _ in fns]
for _ in range(ROUNDS):
    evs = []
    for f, bt_ in zip(fns, between):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        evs.append((e0, e1))
        bt_()
    torch.cuda.synchronize()
    for t, (e0, e1) in zip(times, evs):
        t.append(e0.elapsed_time(e1) * 1e3)
rows, tot_med, tot_min = [], 0.0, 0.0
for n, k, fl, t in zip(names, kernels, flops, times):
    med, mn = statistics.median(t), min(t)
    tot_med += med; tot_min += mn
    rows.append({"layer": n, "kernel": k, "gflop": round(fl / 1e9, 1), "us_median": round(med, 1), "us_min": round(mn, 1),
                 "tflops_median": round(fl / med / 1e6, 1), "frac_of_peak": round(fl / med / 1e6 / PEAK, 4)})
fl = sum(flops)
out = {"what": "Generator transposed-conv stack forward (a2 + a3), B=%d, bf16, 128x128, conv kernels only" % B, "rounds": ROUNDS,
       "method": ("chained: each conv reads the map the (untimed) InstanceNorm + LeakyReLU apply launch has just written, as in Generator.__call__"
                  if CHAINED else "static: every layer on an input tensor of its own (all inputs cold)"),
       "layers": rows, "total_gflop": round(fl / 1e9, 1), "total_us_median": round(tot_med, 1), "total_us_min": round(tot_min, 1),
       "tflops": round(fl / tot_med / 1e6, 1), "frac_of_bf16_peak": round(fl / tot_med / 1e6 / PEAK, 4), "peak_tflops": PEAK,
       "target": "north star: >= 0.40", "device": torch.cuda.get_device_name(0)}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
