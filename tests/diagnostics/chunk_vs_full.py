"""Where does a B-image forward first differ from the same forward on slices?  (per-sample ops: it should not, bit for bit)
usage: python tests/diagnostics/chunk_vs_full.py [init_dim] [B] [chunk]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_oracle as O  # noqa: E402
from test_step_gpu import build, dev_inputs, f32_round, perturbed  # noqa: E402

init_dim, B, chunk = (int(a) for a in (sys.argv[1:4] + ["16", "64", "16"][len(sys.argv) - 1:]))
cfg = O.Cfg(init_dim=init_dim, cond_dim=40, batch_size=B)
W = perturbed(cfg, 7)
tr = build(cfg, W, "bf16")
inp = dev_inputs(f32_round(O.make_inputs(cfg, B, seed=21)))


def fwd(lo, hi):
    cg, cd = {}, {}
    fake = tr.generator([inp["noise"][lo:hi].contiguous(), inp["real_cond_2"][lo:hi].contiguous()], cg)
    p = tr.discriminator.forward_packed(fake, cd)
    t = {"dn.u": cg["dn"][1], "dn.st": cg["dn"][2]}
    for i, (x, z, st, x16) in enumerate(cg["dec"], 1):
        t[f"dec{i}.x16"], t[f"dec{i}.z"], t[f"dec{i}.st"] = x16, z, st
    t["xdec16"], t["fake"] = cg["xdec16"], fake
    for i, (x, z, st, x16) in enumerate(cd["enc"], 1):
        t[f"enc{i}.z"], t[f"enc{i}.st"] = z, st
    t["heads_x"], t["p"] = cd["heads_x"], p
    return {k: v.clone() for k, v in t.items() if v is not None}


full = fwd(0, B)
for lo in range(0, B, chunk):
    part = fwd(lo, lo + chunk)
    line = []
    for k, v in part.items():
        f = full[k][lo:lo + chunk]
        if not torch.equal(v, f):
            d = (v.float() - f.float()).abs()
            line.append(f"{k}: {int((d > 0).sum())}/{d.numel()} differ, max {float(d.max()):.3e}")
    print(f"rows {lo}..{lo + chunk}: " + ("identical" if not line else "; ".join(line[:6])))
