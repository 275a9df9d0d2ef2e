"""Diagnostic (GPU): A tape. Record HIP g_adj and decoder-output gradient; replay through the fp64 oracle."""
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
    o_l1(t, img, g_in, dpre, loss, lam, acc); torch.cuda.synchronize()
    cap.setdefault("l1", []).append((g_in.clone(), dpre.clone(), img.clone()))
ops.l1_tanh_loss = rec_l1
dec = tr.generator.decoder
origd = dec.backward
def hookd(ctx, g_h, need_wgrad):
    out = origd(ctx, g_h, need_wgrad); torch.cuda.synchronize()
    cap.setdefault("dec", []).append((g_h.clone(), out.clone(), need_wgrad))
    return out
dec.backward = hookd
tr.train_step_from_inputs(11, dev_inputs(inp))
gA = grads_of(tr, "A")
ref = O.step_gradients(cfg, W, 11, inp)
# oracle A tape pieces
img1, c1, img2, c2 = inp["real_image_1"], inp["real_cond_1"], inp["real_image_2"], inp["real_cond_2"]
fake = ref["fake_image"]
adj_in_cond = (np.concatenate([c2, c1], 0) + 1.0) * 0.5
adj_t_cond = np.concatenate([c2, c1], 0)
adj_in_img = np.concatenate([img1, fake], 0)
adj_t_img = np.concatenate([img2, img1], 0)
adj_img, acache = O.adjuster_fwd(cfg, W, adj_in_img, adj_in_cond)
(apr, ac), dcache = O.discriminator_fwd(cfg, W["D"], adj_img)
_, g_adj = O.discriminator_bwd(cfg, W["D"], dcache, O.bce_mean_bwd(O.soft(1.0), apr), O.bce_mean_bwd(adj_t_cond, ac), need_wgrad=False, need_input_grad=True)
l1g = cfg.l1_lambda * O.l1_mean_bwd_b(adj_t_img, adj_img)
def relv(got, exp):
    exp = np.asarray(exp, np.float64).ravel(); got = np.asarray(got, np.float64).ravel()[:exp.size]
    return np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30), np.sqrt(((got-exp)**2).mean())/np.sqrt((exp**2).mean())
g_adj_hip, dpre_hip, img_hip = [t.cpu().double().numpy() for t in cap["l1"][1]]
print("adj_img err", relv(img_hip, adj_img), " g_adj err", relv(g_adj_hip, g_adj))
dA_ref = O.adjuster_bwd_own(cfg, W, acache, g_adj + l1g)
dA_mix = O.adjuster_bwd_own(cfg, W, acache, g_adj_hip + l1g)
print("oracleA(HIP g_adj) vs oracle:", [f"{relv(a,b)[1]:.1e}" for a, b in zip(dA_mix, dA_ref)])
print("HIP dA vs oracle            :", [f"{relv(a,b)[1]:.1e}" for a, b in zip(gA, dA_ref)])
# decoder-output gradient in the A tape (2nd decoder.backward call)
g_in_hip, g_out_hip, nw = cap["dec"][1]
cond, u, nc, dcaches, img = acache
dpre = (g_adj + l1g) * (1 - img * img)
dxdec = O.conv_fwd(dpre, W["G"][20], 1)
_, dw4 = O.decoder_bwd(cfg, W["G"][4:20], dcaches, dxdec, need_wgrad=False)
print("decoder in-grad err", relv(g_in_hip.cpu().double().numpy(), dxdec), " decoder out-grad err", relv(g_out_hip.cpu().double().numpy(), dw4))
dv, dg, dbe = O.instnorm_bwd(nc, W["A"][2][0], dw4.reshape(u.shape))
dv2, dg2, dbe2 = O.instnorm_bwd(nc, W["A"][2][0], g_out_hip.cpu().double().numpy().reshape(u.shape))
print("A norm bwd on HIP out-grad: dgamma", dg2, "oracle", dg, " dv err", relv(dv2, dv))
print("|dw4| rms", np.sqrt((dw4**2).mean()), "|dv| rms", np.sqrt((dv**2).mean()))
