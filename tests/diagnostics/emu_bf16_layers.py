"""Diagnostic: per-layer comparison of the bf16 HIP forward/backward against the bf16-emulating oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import np_oracle as O
from test_step_gpu import build, dev_inputs, f32_round, perturbed, grads_of


def rel(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30)


def run(cfg, seed=1, b=11):
    W = perturbed(cfg, seed)
    tr = build(cfg, W, "bf16")
    inp = f32_round(O.make_inputs(cfg, cfg.batch_size, seed=9))
    cfg_e = O.Cfg(**{**cfg.__dict__, "emulate_bf16": True})
    d = dev_inputs(inp)
    # ---- G forward
    ctx = {}
    fake = tr.generator([d["noise"], d["real_cond_2"]], ctx)
    img_ref, gc = O.generator_fwd(cfg_e, W["G"], inp["noise"], inp["real_cond_2"])
    x0, u, nc, dcaches, xdec, img = gc
    print("G: fake", rel(fake.cpu().numpy(), img_ref))
    for i, (x, z, st, x16) in enumerate(ctx["dec"]):
        xr, y, (c, sigma, s) = dcaches[i]
        mu_k = (st[:, 0].double() + st[:, 4].double()).cpu().numpy(); sig_k = st[:, 1].double().cpu().numpy()
        zk = z.float().cpu().numpy().reshape(z.shape[0], -1)
        ck = zk - mu_k[:, None]
        print(f"  dec{i+1}: x16 {rel(x16.float().cpu().numpy(), xr):.2e}  c(z16-mu) {rel(ck, c):.2e}  sigma {rel(sig_k, sigma.ravel()):.2e}")
    # ---- D forward on new_image
    ctx_d = {}
    p = tr.discriminator.forward_packed(d["new_image"], ctx_d)
    (pr, cc), dc = O.discriminator_fwd(cfg_e, W["D"], inp["new_image"])
    ecaches = dc[0]
    print("D: p", rel(p.cpu().numpy()[:, :1], pr), rel(p.cpu().numpy()[:, 1:], cc))
    for i, (x, z, st, x16) in enumerate(ctx_d["enc"]):
        xr, y, (c, sigma, s) = ecaches[i]
        mu_k = (st[:, 0].double() + st[:, 4].double()).cpu().numpy(); sig_k = st[:, 1].double().cpu().numpy()
        zk = z.float().cpu().numpy().reshape(z.shape[0], -1)
        ck = zk - mu_k[:, None]
        xs = x16.float().cpu().numpy() if x16 is not None else x.cpu().numpy()
        print(f"  enc{i+1}: x {rel(xs, xr):.2e}  c(z16-mu) {rel(ck, c):.2e}  sigma {rel(sig_k, sigma.ravel()):.2e}")
    # ---- whole step gradients per tensor, with the backward intermediates of both sides recorded in call order
    from littlegan_amd import ops
    rec = []
    def wrap(name, fn, pick):
        def f(*a, **k):
            out = fn(*a, **k)
            rec.append((name, pick(out, a, k)))
            return out
        return f
    o_bwd, o_c2d, o_ctd, o_s1 = ops.instnorm_bwd, ops.conv2d_s2_dgrad, ops.convT_s2_dgrad, ops.convT_s1_tanh_bwd
    ops.instnorm_bwd = wrap("norm_bwd", o_bwd, lambda out, a, k: (out if out is not None else k.get("out16")).float().cpu().numpy())
    ops.conv2d_s2_dgrad = wrap("dgrad", o_c2d, lambda out, a, k: out.float().cpu().numpy())
    ops.convT_s2_dgrad = wrap("dgrad", o_ctd, lambda out, a, k: out.float().cpu().numpy())
    ops.convT_s1_tanh_bwd = wrap("final", o_s1, lambda out, a, k: (a[1].float().cpu().numpy(), out.float().cpu().numpy() if out is not None else None))
    O.TRACE = []
    refe = O.step_gradients(cfg_e, W, b, inp)
    tr.train_step_from_inputs(b, d)
    torch.cuda.synchronize()
    ops.instnorm_bwd, ops.conv2d_s2_dgrad, ops.convT_s2_dgrad, ops.convT_s1_tanh_bwd = o_bwd, o_c2d, o_ctd, o_s1
    # oracle order (disc tape real: enc4..1 ; disc tape fake: enc4..1 ; gen tape: enc4..1, final, dec4..1 ; adj ...)
    print("   kernel calls:", [n for n, _ in rec][:60])
    print("   oracle trace:", [n for n, _ in O.TRACE][:80])
    B = cfg.batch_size
    # gen tape on the kernel side: the third D.backward (rows = fake); find it: after the disc-tape sequence of 4 norm_bwd (+3 dgrad)
    kn = [v for n, v in rec if n == "norm_bwd"]
    kd = [v for n, v in rec if n == "dgrad"]
    kf = [v for n, v in rec if n == "final"]
    on = [v for n, v in O.TRACE if n.endswith(".dz")]
    od = [v for n, v in O.TRACE if n.endswith(".dx") and not n.startswith("final")]
    # disc tape (kernel: one pass on 2B rows; oracle: real then fake)
    for lvl in range(4):
        k = kn[lvl]
        o = np.concatenate([on[lvl], on[4 + lvl]], 0)
        print(f"   disc tape enc{4 - lvl}.dz {rel(k, o):.2e}   real rows {rel(k[:B], on[lvl]):.2e}  fake rows {rel(k[B:], on[4 + lvl]):.2e}")
    for lvl in range(4):
        print(f"   gen tape  enc{4 - lvl}.dz {rel(kn[4 + lvl], on[8 + lvl]):.2e}")
    gd = [v for n, v in O.TRACE if n.startswith("enc") and n.endswith(".dx")]
    print("   n kernel dgrads", len(kd), "oracle enc dx", len(gd))
    dp, dxk = kf[0]
    of = dict((n, v) for n, v in O.TRACE if n.startswith("final"))
    print(f"   final dpre {rel(dp, of['final.dpre']):.2e} dx {rel(dxk, of['final.dx']):.2e}")
    for lvl in range(4):
        print(f"   gen tape  dec{4 - lvl}.dz {rel(kn[8 + lvl], on[12 + lvl]):.2e}")
    O.TRACE = None
    names = O.weight_shapes(cfg)
    for m, key in (("D", "dD"), ("G", "dG"), ("A", "dA")):
        for (nm, shp), got, exp in zip(names[m], grads_of(tr, m), refe[key]):
            exp = np.asarray(exp).ravel()
            print(f"   {nm:24s} {rel(got[:exp.size], exp):.2e}  |exp|max {np.abs(exp).max():.2e}")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "a"
    if which == "a":
        run(O.Cfg(init_dim=2, conv_filter=(64, 32, 32, 64, 32), cond_dim=5, noise_dim=11, batch_size=3))
    elif which == "b":
        run(O.Cfg(init_dim=2, conv_filter=(64, 32, 32, 32, 32), cond_dim=5, noise_dim=11, batch_size=3))
    else:
        run(O.Cfg(init_dim=int(which), cond_dim=40, batch_size=2), seed=7)
