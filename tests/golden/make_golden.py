"""Generates tests/golden/step_small.npz from the fp64 numpy oracle.

The reference (TF-1.15) cannot run in this pipeline and ships no fixtures
(SURVEY.md §8c), so these vectors pin the ORACLE's own outputs (cross-checked
against torch autograd in tests/test_oracle.py), not TensorFlow's: parity unpinned.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_oracle as O  # noqa: E402

CFG = dict(init_dim=2, conv_filter=(64, 32, 32, 32, 32), cond_dim=5, noise_dim=11, batch_size=2)
STEPS = (9, 10, 11, 12)  # 10 = partition step without Adjuster, 11/12 = Adjuster branch on


def main():
    cfg = O.Cfg(**CFG)
    W = O.init_weights(cfg, seed=3)
    rng = np.random.default_rng(11)
    for m in W:  # non-trivial biases / gamma / beta
        for i, w in enumerate(W[m]):
            if w.ndim == 1:
                W[m][i] = w + 0.05 * rng.standard_normal(w.shape)
    # the run STARTS from the float32-rounded values that are stored as W0
    W = {m: [w.astype(np.float32).astype(np.float64) for w in ws] for m, ws in W.items()}
    st = O.TrainState(cfg, {m: [w.copy() for w in ws] for m, ws in W.items()})
    save = {}
    for m in W:
        for i, w in enumerate(W[m]):
            save[f"W0_{m}_{i}"] = w.astype(np.float32)
    for b in STEPS:
        inp = O.make_inputs(cfg, cfg.batch_size, seed=100 + b)
        inp = {k: v.astype(np.float32).astype(np.float64) for k, v in inp.items()}
        out = O.train_step(st, b, inp)
        for k, v in inp.items():
            save[f"in{b}_{k}"] = v.astype(np.float32)
        save[f"out{b}_fake_image"] = out["fake_image"].astype(np.float32)
        save[f"out{b}_gen_loss"] = np.float64(out["gen_loss"])
        save[f"out{b}_disc_loss"] = np.float64(out["disc_loss"])
        if out["adj_image"] is not None:
            save[f"out{b}_adj_image"] = out["adj_image"].astype(np.float32)
            save[f"out{b}_adj_loss"] = np.float64(out["adj_loss"])
    for m in st.W:
        for i, w in enumerate(st.W[m]):
            save[f"W4_{m}_{i}"] = w.astype(np.float32)  # expected values, rounded for storage
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "step_small.npz"), **save)
    print("wrote step_small.npz", sum(v.nbytes for v in save.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    main()
