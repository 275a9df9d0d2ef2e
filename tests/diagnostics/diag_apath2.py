"""Diagnostic (GPU): stage-local errors of every instnorm_bwd / convT_s2_dgrad call of the A tape."""
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
rec = []
o_in, o_dg = ops.instnorm_bwd, ops.convT_s2_dgrad
def rec_in(x, stats, g, dgamma, dbeta, pre, post, alpha, accumulate=False, out=None):
    r = o_in(x, stats, g, dgamma, dbeta, pre, post, alpha, accumulate, out); torch.cuda.synchronize()
    rec.append(("norm", x.clone(), stats.clone(), g.clone(), r.clone(), pre, post)); return r
def rec_dg(dy, pack, cs, dtype, out=None):
    r = o_dg(dy, pack, cs, dtype, out); torch.cuda.synchronize()
    rec.append(("dgrad", dy.clone(), r.clone(), cs)); return r
A = tr.adjuster
origA = A.backward_own
def hookA(ctx, dpre):
    ops.instnorm_bwd, ops.convT_s2_dgrad = rec_in, rec_dg
    import littlegan_amd.model as M
    origA(ctx, dpre)
    ops.instnorm_bwd, ops.convT_s2_dgrad = o_in, o_dg
A.backward_own = hookA
Wg = {i: w.copy() for i, w in enumerate(W["G"])}
tr.train_step_from_inputs(11, dev_inputs(inp))
def relv(got, exp):
    exp = np.asarray(exp, np.float64).ravel(); got = np.asarray(got, np.float64).ravel()
    return np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30), np.sqrt(((got-exp)**2).mean())/np.sqrt((exp**2).mean())
lvl = 4
for r in rec:
    if r[0] == "norm":
        _, x, st, g, out, pre, post = r
        xn, gn = x.cpu().double().numpy(), g.cpu().double().numpy().reshape(x.shape)
        if pre:   # A._dn
            gamma, beta = W["A"][2][0], W["A"][3][0]
            xx = O.leaky(xn, 0.3); y, cache = O.instnorm(xx, gamma, beta)
            d, _, _ = O.instnorm_bwd(cache, gamma, gn); d = O.leaky_bwd(xn, d, 0.3)
        else:
            gamma, beta = W["G"][4 + 4 * (lvl - 1) + 2][0], W["G"][4 + 4 * (lvl - 1) + 3][0]
            y, cache = O.instnorm(xn, gamma, beta)
            d, _, _ = O.instnorm_bwd(cache, gamma, O.leaky_bwd(y, gn, 0.3))
        B = xn.shape[0]
        c, sigma, s = cache
        print("norm  lvl", lvl, tuple(x.shape), "local err", relv(out.cpu().numpy(), d), " |g| rms %.3e |out| rms %.3e" % (np.sqrt((gn**2).mean()), np.sqrt((d**2).mean())),
              " sum(out)/sum|out| per sample", [float(out[i].double().sum() / out[i].double().abs().sum()) for i in range(B)])
    else:
        _, dy, out, cs = r
        k = W["G"][4 + 4 * (lvl - 1)]
        d = O.conv_fwd(dy.cpu().double().numpy(), k, 2)
        print("dgrad lvl", lvl, tuple(dy.shape), "local err", relv(out.cpu().numpy(), d))
        lvl -= 1

# ---- cumulative errors vs the full fp64 oracle chain
ref = O.step_gradients(cfg, W, 11, inp)
img1, c1, img2, c2 = inp["real_image_1"], inp["real_cond_1"], inp["real_image_2"], inp["real_cond_2"]
fake = ref["fake_image"]
adj_in_cond = (np.concatenate([c2, c1], 0) + 1.0) * 0.5
adj_t_cond = np.concatenate([c2, c1], 0)
adj_in_img = np.concatenate([img1, fake], 0)
adj_t_img = np.concatenate([img2, img1], 0)
adj_img, acache = O.adjuster_fwd(cfg, W, adj_in_img, adj_in_cond)
(apr, ac), dcache = O.discriminator_fwd(cfg, W["D"], adj_img)
_, g_adj = O.discriminator_bwd(cfg, W["D"], dcache, O.bce_mean_bwd(O.soft(1.0), apr), O.bce_mean_bwd(adj_t_cond, ac), need_wgrad=False, need_input_grad=True)
cond, u, nc, dcaches, img = acache
dpre = (g_adj + cfg.l1_lambda * O.l1_mean_bwd_b(adj_t_img, adj_img)) * (1 - img * img)
g_h = O.conv_fwd(dpre, W["G"][20], 1)
it = iter(rec)
for i in reversed(range(4)):
    k, b, g, be = W["G"][4 + 4 * i:8 + 4 * i]
    x, y, ncache = dcaches[i]
    r = next(it)
    print("cum lvl", i + 1, "norm in-grad err", relv(r[3].cpu().numpy(), g_h), "leaky-mask mismatches", int(((r[1].cpu().double().numpy()*0+y > 0) != ((r[2][:, 2].cpu().double().numpy().reshape(-1,1,1,1) * (r[1].cpu().double().numpy() - r[2][:, 0].cpu().double().numpy().reshape(-1,1,1,1) - r[2][:, 4].cpu().double().numpy().reshape(-1,1,1,1)) + r[2][:, 3].cpu().double().numpy().reshape(-1,1,1,1)) > 0)).sum()))
    dy = O.leaky_bwd(y, g_h, 0.3)
    dz, _, _ = O.instnorm_bwd(ncache, g[0], dy)
    print("           norm out err", relv(r[4].cpu().numpy(), dz))
    r = next(it)
    g_h = O.conv_fwd(dz, k, 2)
    print("           dgrad out err", relv(r[2].cpu().numpy(), g_h))
