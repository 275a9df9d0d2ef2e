"""Diagnostic (GPU): where does the G-tape error enter?  Captures dpre and the final-conv dx."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import np_oracle as O
from tests.test_step_gpu import build, dev_inputs, f32_round, perturbed

cfg = O.Cfg(init_dim=4, cond_dim=40, batch_size=2)
W = perturbed(cfg, 7)
tr = build(cfg, W, "f32")
inp = f32_round(O.make_inputs(cfg, 2, seed=9))
cap = {}
conv = tr.generator.conv
orig = conv.backward
def hooked(x, dpre, need_wgrad):
    dx = orig(x, dpre, need_wgrad)
    torch.cuda.synchronize()
    cap.setdefault("calls", []).append((None if x is None else x.clone(), dpre.clone(), dx.clone(), need_wgrad))
    return dx
conv.backward = hooked
dec = tr.generator.decoder
origd = dec.backward
def hookd(ctx, g_h, need_wgrad):
    out = origd(ctx, g_h, need_wgrad)
    torch.cuda.synchronize()
    cap.setdefault("dec", []).append((out.clone(), need_wgrad, [tuple(t.clone() for t in lvl) for lvl in ctx["dec"]]))
    return out
dec.backward = hookd
tr.train_step_from_inputs(11, dev_inputs(inp))

# oracle intermediates for the G tape
Wg, Wd = W["G"], W["D"]
fake, gcache = O.generator_fwd(cfg, Wg, inp["noise"], inp["real_cond_2"])
(fpr, fc), fcache = O.discriminator_fwd(cfg, Wd, fake)
_, d_fake = O.discriminator_bwd(cfg, Wd, fcache, O.bce_mean_bwd(O.soft(1.0), fpr), O.bce_mean_bwd(inp["real_cond_2"], fc), need_wgrad=False, need_input_grad=True)
d_fake = d_fake + cfg.l1_lambda * O.l1_mean_bwd_b(inp["real_image_2"], fake)
x0, u, nc, dcaches, xdec, img = gcache
dpre = d_fake * (1 - img * img)
dxdec = O.conv_fwd(dpre, Wg[20], 1)
def rel(a, b):
    a = a.detach().cpu().double().numpy()
    return np.abs(a - b).max() / np.abs(b).max(), np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean())
x, dp, dx, nw = cap["calls"][0]
print("G tape: need_wgrad", nw, "dpre err", rel(dp, dpre), "xdec err", rel(x, xdec), "dx err", rel(dx, dxdec))
# recompute dx standalone from the captured dpre with the same kernel
from littlegan_amd import ops
dx2 = torch.empty_like(dx)
ops.convT_s1_tanh_bwd(None, dp, conv.pack(), conv.cs, conv.dtype, dx=dx2)
print("standalone dx err", rel(dx2, dxdec), "dx vs dx2", (dx - dx2).abs().max().item())
# saved decoder context vs oracle
g0, nw, lv = cap["dec"][0]
for i, (xs, zs, st) in enumerate(lv):
    xo, yo, nco = dcaches[i]
    print(" level", i + 1, "x err", rel(xs, xo), "stats mu err", np.abs(st[:, 0].cpu().numpy() - (xo * 0 + 0).mean()) .max() if False else "", )
    zo = O.conv2d_transpose(xo, Wg[4 + 4 * i], Wg[5 + 4 * i], 2)
    print("          z err", rel(zs, zo))
