"""Diagnostic (not a test): error of the bf16 HIP step against the bf16-EMULATING oracle and against the plain fp64 oracle.
python tests/diagnostics/emu_bf16_report.py [init_dim ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import np_oracle as O
from test_step_gpu import build, dev_inputs, f32_round, perturbed, grads_of


def report(cfg, tag, seed=7, b=11):
    W = perturbed(cfg, seed)
    tr = build(cfg, W, "bf16")
    inp = f32_round(O.make_inputs(cfg, cfg.batch_size, seed=9))
    t0 = time.time()
    ref64 = O.step_gradients(cfg, W, b, inp)
    cfg_e = O.Cfg(**{**cfg.__dict__, "emulate_bf16": True})
    fake, adj, lg, ld, la = tr.train_step_from_inputs(b, dev_inputs(inp))
    torch.cuda.synchronize()
    refe = O.step_gradients(cfg_e, W, b, inp, fake_override=fake.cpu().numpy(), adj_override=adj.cpu().numpy() if adj is not None else None)
    print("  own-image check: fake", np.abs(fake.cpu().numpy() - refe["fake_image_own"]).max(),
          "adj", np.abs(adj.cpu().numpy() - refe["adj_image_own"]).max() if adj is not None else None)
    print(f"== {tag}  (oracle {time.time() - t0:.1f}s)")
    for name, ref in (("emu", refe), ("f64", ref64)):
        e_img = np.abs(fake.cpu().numpy() - ref["fake_image"]).max()
        e_adj = np.abs(adj.cpu().numpy() - ref["adj_image"]).max() if adj is not None else 0
        el = [abs(g.item() - ref[k]) / abs(ref[k]) for g, k in ((lg, "gen_loss"), (ld, "disc_loss"), (la, "adj_loss")) if g is not None]
        print(f"  vs {name}: img {e_img:.2e} adj {e_adj:.2e} losses {['%.1e' % v for v in el]}")
        for m, key in (("D", "dD"), ("G", "dG"), ("A", "dA")):
            if ref[key] is None:
                continue
            exps = [np.asarray(e, np.float64).ravel() for e in ref[key]]
            gmax = max(np.abs(e).max() for e in exps)
            rms, sc = [], []
            for got, exp in zip(grads_of(tr, m), exps):
                d = got[:exp.size] - exp
                if exp.size == 1:
                    sc.append(abs(d[0]) / gmax)
                else:
                    rms.append(np.sqrt((d * d).mean()) / (np.sqrt((exp * exp).mean()) + 1e-30))
            print(f"     {m}: tensor rms max {max(rms):.2e} median {np.median(rms):.2e} | scalars |d|/gmax max {max(sc) if sc else 0:.2e}")


if __name__ == "__main__":
    dims = [int(v) for v in sys.argv[1:]] or [4, 8]
    report(O.Cfg(init_dim=2, conv_filter=(64, 32, 32, 64, 32), cond_dim=5, noise_dim=11, batch_size=3), "small 32x32 B=3", seed=1)
    report(O.Cfg(init_dim=2, conv_filter=(64, 32, 32, 32, 32), cond_dim=5, noise_dim=11, batch_size=3), "small (conv1 N=32) B=3", seed=1)
    for d in dims:
        report(O.Cfg(init_dim=d, cond_dim=40, batch_size=2), f"full channels {16 * d}x{16 * d} B=2")
