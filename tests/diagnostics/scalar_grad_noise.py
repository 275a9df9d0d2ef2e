"""How far do the 1-element gradient tensors (gamma / beta: sums with heavy cancellation) of one bf16 step sit from the bf16-emulating
oracle, and how much does that distance move between EQUIVALENT kernel choices?  Run once per environment (the switches are read once per
process), e.g.  LG_ROWS_XJ1=1 / LG_NO_ZN=1 / LG_NO_UP4_PAIR=1:  the kernels behind those switches compute the same sums in another
order, so a difference between the runs is the noise floor of the whole-step comparison, not an error of either kernel.
Measured in round 3 (init_dim 8): between the build whose final layer added its taps in kx order 0..4 and the one adding them 4..0
(same products, fp32 sums in another order: images differ in the last bit) the ORACLE's Adjuster norm-beta gradient moved from 0.011329
to 0.009519 = 0.11 of the model's largest gradient, the kernels' value from 0.011319 to 0.011492: bf16 rounding flips downstream of a
1e-7 change decide such a cancelling sum.  Hence sfloor = 0.22 in test_step_gpu.TOLS["bf16_emu"].
  python tests/diagnostics/scalar_grad_noise.py [init_dim]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import np_oracle as O
from test_step_gpu import build, dev_inputs, emu_reference, f32_round, grads_of, perturbed, TOLS

init_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = O.Cfg(init_dim=init_dim, cond_dim=40, batch_size=2)
W = perturbed(cfg, 7)
tr = build(cfg, W, "bf16")
inp = f32_round(O.make_inputs(cfg, 2, seed=9))
fake, adj, lg, ld, la = tr.train_step_from_inputs(11, dev_inputs(inp))
ref = emu_reference(cfg, W, 11, inp, fake, adj)
tol = TOLS["bf16_emu"]
sw = {k: v for k, v in os.environ.items() if k.startswith("LG_")}
print("switches:", sw)
for m, key in (("D", "dD"), ("G", "dG"), ("A", "dA")):
    exps = [np.asarray(e, np.float64).ravel() for e in ref[key]]
    gmax = max(np.abs(e).max() for e in exps)
    for i, (got, exp) in enumerate(zip(grads_of(tr, m), exps)):
        if exp.size == 1:
            d = got[0] - exp[0]
            bound = tol["grad_rms"] * abs(exp[0]) + tol["sfloor"] * gmax
            print(f"  {m}[{i}] got {got[0]:+.6f} oracle {exp[0]:+.6f} diff {d:+.6f} = {abs(d) / bound:5.2f} of the bound ({abs(d) / gmax:.3f} of the model's largest gradient)")
