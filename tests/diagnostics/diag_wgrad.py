import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import np_oracle as O
from littlegan_amd import ops
def dev(a): return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
def rel(got, exp):
    got = got.detach().cpu().double().numpy()
    return np.abs(got - exp).max() / (np.abs(exp).max() + 1e-30)
rng = np.random.default_rng(0)
for (B, Hs, Ws, cb, cs) in [(2, 32, 32, 32, 64), (2, 16, 16, 64, 128), (4, 32, 32, 32, 64), (2, 32, 32, 32, 32), (1, 64, 64, 32, 64), (2, 8, 8, 128, 256)]:
    x = rng.standard_normal((B, Hs, Ws, cs)).astype(np.float32).astype(np.float64)
    dy = rng.standard_normal((B, 2 * Hs, 2 * Ws, cb)).astype(np.float32).astype(np.float64)
    dw_e = O.conv_bwd_filter(dy, x, 2, 5)
    for dt in (0, 1):
        dw = torch.zeros(5, 5, cb, cs, device="cuda")
        ops.convT_s2_wgrad(dev(x), dev(dy), dw, False, dt)
        e = np.abs(dw.cpu().double().numpy() - dw_e)
        print((B, Hs, Ws, cb, cs), "dtype", dt, "wgrad rel err", rel(dw, dw_e), "argmax tap", np.unravel_index(e.argmax(), e.shape))
    db = torch.empty(cb, device="cuda")
    ops.bias_grad(dev(dy), db)
    print("   bias_grad rel err", rel(db, dy.sum((0, 1, 2))), "M", B * 4 * Hs * Ws)
