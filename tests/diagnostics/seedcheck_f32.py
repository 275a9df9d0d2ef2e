"""Gradient error of one f32 step at 128x128 (full channels, B=2) against the fp64 oracle for several input seeds:
shows the LeakyReLU sign-flip sensitivity the tolerances of tests/test_step_gpu.py account for."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import np_oracle as O
from test_step_gpu import TOLS, build, dev_inputs, f32_round, perturbed
cfg = O.Cfg(init_dim=8, cond_dim=40, batch_size=2)
for seed in (9, 10, 11, 12):
    W = perturbed(cfg, 7)
    inp = f32_round(O.make_inputs(cfg, 2, seed=seed))
    ref = O.step_gradients(cfg, W, 11, inp)
    tr = build(cfg, W, "f32")
    tr.train_step_from_inputs(11, dev_inputs(inp))
    errs = []
    for m, key in (("D", "dD"), ("G", "dG"), ("A", "dA")):
        s, e = tr.store.model_range(m)
        for (a, b), exp in zip(tr.store.ranges[m], ref[key]):
            got = tr.store.grad[a:a + exp.size].cpu().numpy().reshape(exp.shape)
            if exp.size > 1:
                errs.append(np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30))
    print(seed, "median", np.median(errs), "max", max(errs), flush=True)
