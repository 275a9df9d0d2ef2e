"""One full-channel step at 256x256 (C5 layer geometry, init_dim=16, B=2) against the fp64 oracle: images, losses and
gradients with the tolerances of tests/test_step_gpu.py.  Manual check (the oracle step takes ~1 min of CPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from oracle import np_oracle as O
from test_step_gpu import TOLS, build, check_grads, dev_inputs, f32_round, perturbed

cfg = O.Cfg(init_dim=16, cond_dim=40, batch_size=2)
W = perturbed(cfg, 7)
inp = f32_round(O.make_inputs(cfg, 2, seed=9))
t = time.time(); ref = O.step_gradients(cfg, W, 11, inp); print("oracle %.1f s" % (time.time() - t), flush=True)
for mfma in ("f32", "bf16"):
    tol = dict(TOLS[mfma])
    if mfma == "f32":  # 4x more elements per map than the 128x128 test: more LeakyReLU pre-activations within fp32 rounding of
        tol["grad_med"] *= 8  # zero flip sign (DESIGN.md par. 2); measured median 9e-5, images and losses stay at 1e-5 / 1e-6
    tr = build(cfg, W, mfma)
    fake, adj, lg, ld, la = tr.train_step_from_inputs(11, dev_inputs(inp))
    e1 = np.abs(fake.cpu().numpy() - ref["fake_image"]).max(); e2 = np.abs(adj.cpu().numpy() - ref["adj_image"]).max()
    print(mfma, "img err", e1, e2, "losses", lg.item(), ref["gen_loss"], ld.item(), ref["disc_loss"], la.item(), ref["adj_loss"], flush=True)
    assert e1 < tol["img"] and e2 < tol["img"]
    for got, key in ((lg, "gen_loss"), (ld, "disc_loss"), (la, "adj_loss")):
        assert abs(got.item() - ref[key]) < tol["loss"] * abs(ref[key])
    check_grads(tr, ref, (("D", "dD"), ("G", "dG"), ("A", "dA")), tol)
    print(mfma, "256x256 step OK", flush=True)
