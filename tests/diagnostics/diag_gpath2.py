"""Diagnostic (GPU): G tape, level-4 dz and the gradients right after G.backward vs after the whole step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import np_oracle as O
from tests.test_step_gpu import build, dev_inputs, f32_round, perturbed, grads_of
from littlegan_amd import ops

cfg = O.Cfg(init_dim=4, cond_dim=40, batch_size=2)
W = perturbed(cfg, 7)
tr = build(cfg, W, "f32")
inp = f32_round(O.make_inputs(cfg, 2, seed=9))
ref = O.step_gradients(cfg, W, 11, inp)
cap = {}
G = tr.generator
origG = G.backward
def hookG(ctx, dpre):
    rec = []
    o_in = ops.instnorm_bwd
    def rec_in(x, stats, g, dgamma, dbeta, pre, post, alpha, accumulate=False, out=None):
        r = o_in(x, stats, g, dgamma, dbeta, pre, post, alpha, accumulate, out)
        torch.cuda.synchronize()
        rec.append((x.clone(), stats.clone(), g.clone(), r.clone()))
        return r
    ops.instnorm_bwd = rec_in
    origG(ctx, dpre)
    ops.instnorm_bwd = o_in
    torch.cuda.synchronize()
    cap["rec"] = rec
    cap["gG"] = grads_of(tr, "G")
G.backward = hookG
tr.train_step_from_inputs(11, dev_inputs(inp))
def relv(got, exp):
    exp = np.asarray(exp, np.float64).ravel(); got = np.asarray(got, np.float64).ravel()[:exp.size]
    return np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30)
print("right after G.backward:", " ".join(f"{relv(g, e):.1e}" for g, e in zip(cap["gG"], ref["dG"])))
print("after whole step      :", " ".join(f"{relv(g, e):.1e}" for g, e in zip(grads_of(tr, "G"), ref["dG"])))
# level-4 norm backward (first recorded call) vs oracle computed from the recorded inputs
x, st, g, dz = cap["rec"][0]
xn, gn = x.cpu().double().numpy(), g.cpu().double().numpy()
y, cache = O.instnorm(xn, float(W["G"][18][0]), float(W["G"][19][0]))
dy = O.leaky_bwd(y, gn, 0.3)
dz_ref, dg_ref, db_ref = O.instnorm_bwd(cache, float(W["G"][18][0]), dy)
print("level4 dz err (inputs as recorded):", relv(dz.cpu().numpy(), dz_ref), "shape", tuple(x.shape))
print("stats mu/sigma err", np.abs(st[:, 0].cpu().numpy() - xn.reshape(2, -1).mean(1)).max(), np.abs(st[:, 1].cpu().numpy() - xn.reshape(2, -1).std(1)).max())
print("dgamma", cap["gG"][18][0], dg_ref, "dbeta", cap["gG"][19][0], db_ref)
