"""Diagnostic (GPU): per-tensor relative errors of the HIP step vs the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import np_oracle as O
from tests.test_step_gpu import build, dev_inputs, f32_round, perturbed, grads_of

def run(cfg, mfma, b, seed):
    W = perturbed(cfg, 7)
    tr = build(cfg, W, mfma)
    inp = f32_round(O.make_inputs(cfg, cfg.batch_size, seed=seed))
    ref = O.step_gradients(cfg, W, b, inp)
    fake, adj, lg, ld, la = tr.train_step_from_inputs(b, dev_inputs(inp))
    print(f"== {mfma} init_dim={cfg.init_dim} B={cfg.batch_size} b={b}")
    print(" img", np.abs(fake.cpu().numpy() - ref["fake_image"]).max(), "adj", np.abs(adj.cpu().numpy() - ref["adj_image"]).max())
    for got, key in ((lg, "gen_loss"), (ld, "disc_loss"), (la, "adj_loss")):
        print(" ", key, abs(got.item() - ref[key]) / abs(ref[key]))
    for m, key in (("D", "dD"), ("G", "dG"), ("A", "dA")):
        errs = []
        for got, exp in zip(grads_of(tr, m), ref[key]):
            exp = np.asarray(exp, np.float64).ravel()
            d = np.abs(got[:exp.size] - exp)
            errs.append((d.max() / (np.abs(exp).max() + 1e-30), np.sqrt((d**2).mean()) / (np.sqrt((exp**2).mean()) + 1e-30)))
        print(" ", key, "max:", " ".join(f"{e[0]:.1e}" for e in errs))
        print(" ", key, "rms:", " ".join(f"{e[1]:.1e}" for e in errs))

for mfma in ("f32", "bf16"):
    run(O.Cfg(init_dim=4, cond_dim=40, batch_size=2), mfma, 11, 9)
    run(O.Cfg(init_dim=2, conv_filter=(64, 32, 32, 32, 32), cond_dim=5, noise_dim=11, batch_size=3), mfma, 11, 61)
