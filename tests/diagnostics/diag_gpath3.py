"""Diagnostic (GPU): feed the HIP D-input-gradient into the fp64 oracle's G backward; analyse its error."""
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
cap = {}
o_l1 = ops.l1_tanh_loss
def rec_l1(t, img, g_in, dpre, loss, lam, acc):
    o_l1(t, img, g_in, dpre, loss, lam, acc)
    torch.cuda.synchronize()
    cap.setdefault("l1", []).append((g_in.clone(), dpre.clone(), img.clone()))
ops.l1_tanh_loss = rec_l1
import littlegan_amd.eager_trainer as ET
tr.train_step_from_inputs(11, dev_inputs(inp))
gG = grads_of(tr, "G")
g_img_hip, dpre_hip, img_hip = [t.cpu().double().numpy() for t in cap["l1"][0]]
Wg, Wd = W["G"], W["D"]
fake, gcache = O.generator_fwd(cfg, Wg, inp["noise"], inp["real_cond_2"])
(fpr, fc), fcache = O.discriminator_fwd(cfg, Wd, fake)
_, g_img = O.discriminator_bwd(cfg, Wd, fcache, O.bce_mean_bwd(O.soft(1.0), fpr), O.bce_mean_bwd(inp["real_cond_2"], fc), need_wgrad=False, need_input_grad=True)
d = g_img_hip - g_img
print("g_img: max|g|", np.abs(g_img).max(), "rms g", np.sqrt((g_img**2).mean()), "max err", np.abs(d).max(), "rms err", np.sqrt((d**2).mean()))
for n in range(2):
    eps = (d[n] * g_img[n]).sum() / (g_img[n] ** 2).sum()
    print(" sample", n, "scalar fit eps", eps, "residual rms", np.sqrt(((d[n] - eps * g_img[n]) ** 2).mean()), "mean err", d[n].mean(), "mean g", g_img[n].mean())
print(" border err rms", np.sqrt((d[:, [0, -1]] ** 2).mean()), "interior", np.sqrt((d[:, 8:-8, 8:-8] ** 2).mean()))
ref = O.generator_bwd(cfg, Wg, gcache, g_img + cfg.l1_lambda * O.l1_mean_bwd_b(inp["real_image_2"], fake))
mix = O.generator_bwd(cfg, Wg, gcache, g_img_hip + cfg.l1_lambda * O.l1_mean_bwd_b(inp["real_image_2"], fake))
def relv(got, exp):
    exp = np.asarray(exp, np.float64).ravel(); got = np.asarray(got, np.float64).ravel()[:exp.size]
    return np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30)
print("oracleG(HIP g_img) vs oracle :", " ".join(f"{relv(a, b):.1e}" for a, b in zip(mix, ref)))
print("HIP dG vs oracleG(HIP g_img) :", " ".join(f"{relv(a, b):.1e}" for a, b in zip(gG, mix)))
l1 = cfg.l1_lambda * O.l1_mean_bwd_b(inp["real_image_2"], fake)
print("|l1 term|", np.abs(l1).max(), "rms g_img", np.sqrt((g_img**2).mean()))
