"""CPU tests of the host logic of the product path that needs no GPU:
  * ParamStore layout (22 / 20 / 4 weights, reference order, contiguous optimizer + partition ranges),
  * the partition schedule mirror (eager_trainer.py:104-113),
  * the data-parallel gradient exchange (littlegan_amd.dist.GradSync) on a world_size-2 gloo group:
    mean of the per-rank gradients == gradient of the single-process step on the concatenated batch.
Gradients on CPU come from the torch oracle (the HIP kernels cannot run here); only the exchange is under test."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import np_oracle as O
from oracle import torch_oracle as T

CFG = dict(init_dim=2, conv_filter=(32, 32, 32, 32, 32), cond_dim=3, noise_dim=5, batch_size=2)


def _args(cfg):
    d = {k: getattr(cfg, k) for k in ("batch_size", "image_channel", "noise_dim", "init_dim", "conv_filter", "kernel_size",
                                      "leaky_alpha", "l1_lambda", "lr", "beta_1", "beta_2", "use_clip", "clip_range",
                                      "use_partition", "partition_interval", "train_adj", "cond_dim")}
    return SimpleNamespace(**d, mfma_dtype="f32", device="cpu", seed=0, use_gp=False, no_io=True, dropout_rate=0.5)


def _store(cfg):
    from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator, ParamStore
    a = _args(cfg)
    dec, enc = Decoder(a), Encoder(a)
    g = Generator(a, dec)
    d = Discriminator(a, enc)
    adj = Adjuster(a, d, g)
    return ParamStore(g, d, adj), g, d, adj


def test_param_store_layout_matches_reference_order():
    cfg = O.Cfg(**CFG)
    store, g, d, adj = _store(cfg)
    shapes = O.weight_shapes(cfg)
    for m, model_w in (("G", g.weights), ("D", d.weights), ("A", adj.weights[16:20])):
        assert [tuple(w.shape) for w in model_w] == [s for _, s in shapes[m]]
        assert store.names(m) == [n for n, _ in shapes[m]]
        prev_end = store.ranges[m][0][0]
        for (s, e), w in zip(store.ranges[m], model_w):
            assert s == prev_end and s % 4 == 0 and e - s >= w.numel()  # contiguous, 16-byte aligned
            assert w.data_ptr() == store.flat[s:].data_ptr()  # weights ARE views of the flat buffer
            prev_end = e
    assert len(adj.weights) == 38 and len(g.weights) == 22 and len(d.weights) == 20
    # Adjuster shares encoder / decoder / final conv storage with D / G (model.py:119-123)
    assert adj.weights[0].data_ptr() == d.weights[0].data_ptr()
    assert adj.weights[20].data_ptr() == g.weights[4].data_ptr()
    assert adj.weights[36].data_ptr() == g.weights[20].data_ptr()
    # G, D, A ranges are disjoint and ordered
    assert store.model_range("G")[1] <= store.model_range("D")[0] and store.model_range("D")[1] <= store.model_range("A")[0]


def test_partition_schedule_mirror():
    from littlegan_amd.eager_trainer import train_weight_range
    cfg = O.Cfg(**CFG)
    a = _args(cfg)
    for b in range(1, 31):
        for m in "GDA":
            lo, hi = train_weight_range(a, m, b)
            assert list(range(lo, hi)) == O.train_weight_indices(cfg, m, b)
    a.use_partition = False
    assert train_weight_range(a, "G", 5) == (0, 22)


def test_adam_sets_are_disjoint_so_their_order_is_immaterial():
    """The trainer applies the three optimizers in all-reduce launch order (D, G, A: each set right after ITS wait), the
    reference in the order A, D, G (eager_trainer.py:164-168).  The oracle's own TF-v1 Adam on the flat store's ranges: the
    three weight / slot ranges never overlap and each optimizer owns its beta powers, so both orders give identical bits."""
    cfg = O.Cfg(**CFG)
    store, g, d, adj = _store(cfg)
    spans = {m: store.model_range(m) for m in "GDA"}
    for a_, b_ in (("G", "D"), ("D", "A"), ("G", "A")):
        assert spans[a_][1] <= spans[b_][0] or spans[b_][1] <= spans[a_][0]
    rng = np.random.default_rng(3)
    n = store.flat.numel()
    w0, grad = rng.standard_normal(n), rng.standard_normal(n)
    results = []
    for order in (("A", "D", "G"), ("D", "G", "A")):
        w = [w0.copy()]   # ONE flat weight vector, the optimizers write their own slices
        opts = {"G": O.AdamState(cfg.lr, cfg.beta_1, cfg.beta_2, 1), "D": O.AdamState(cfg.lr, cfg.beta_1, cfg.beta_2, 1),
                "A": O.AdamState(cfg.lr, 0.9, 0.999, 1)}
        for step in range(2):
            for m in order:
                s, e = spans[m]
                part = [w[0][s:e]]
                opts[m].apply(part, [0], [np.clip(grad[s:e], -cfg.clip_range, cfg.clip_range) if m == "D" else grad[s:e]])
                w[0][s:e] = part[0]
        results.append((w[0].copy(), [(o.b1p, o.b2p) for o in opts.values()], [o.m[0].copy() for o in opts.values()]))
    assert np.array_equal(results[0][0], results[1][0]) and results[0][1] == results[1][1]
    assert all(np.array_equal(x, y) for x, y in zip(results[0][2], results[1][2]))
    assert not np.array_equal(results[0][0], w0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_q, batch_no=11):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from littlegan_amd.dist import GradSync
        cfg = O.Cfg(**CFG)
        W = O.init_weights(cfg, 5)
        Bg = cfg.batch_size * world
        full = O.make_inputs(cfg, Bg, seed=77)
        shard = {k: torch.tensor(v[rank * cfg.batch_size:(rank + 1) * cfg.batch_size]) for k, v in full.items()}
        out = T.step_gradients(T.Net(cfg, W, torch.float64), 11, shard)   # full gradient sets; the exchange below picks ranges
        store, g, d, adj = _store(cfg)
        store.grad = store.grad.double()
        for m, key in (("G", "dG"), ("D", "dD"), ("A", "dA")):
            for (s, e), t in zip(store.ranges[m], out[key]):
                store.grad[s:s + t.numel()] = t.reshape(-1)
        sync = GradSync("cpu")
        assert sync.enabled and sync.world_size == world
        from littlegan_amd.eager_trainer import train_weight_range
        a = _args(cfg)
        for m in ("D", "G", "A"):  # launch order of the step: D, G, A; a partition step exchanges its trained range only
            sync.launch(m, store, *store.model_range(m, *train_weight_range(a, m, batch_no)))
        for m in ("D", "G", "A"):   # the step waits per set, right before that set's Adam (eager_trainer.py of this package)
            sync.wait(m)
        assert not sync._pending
        if rank == 0:
            out_q.put(store.grad.numpy().copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch_no", [(2, 11), (4, 11), (4, 15), (4, 20)])
def test_gradsync_ranks_equal_global_batch(world, batch_no):
    """world 2 / 4; batch_no 11 = a full step (three whole-model all-reduces), 15 / 20 = partition steps (group 0 / group 1 of G
    and D: a range-limited all-reduce; the Adjuster has one group = always its whole range, eager_trainer.py:48-52,104-113)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, batch_no)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = O.Cfg(**CFG)
    W = O.init_weights(cfg, 5)
    cfg_g = O.Cfg(**{**CFG, "batch_size": cfg.batch_size * world})
    full = {k: torch.tensor(v) for k, v in O.make_inputs(cfg, cfg.batch_size * world, seed=77).items()}
    # the Adjuster branch concatenates along the batch, so the global-batch step must pair samples like the shards do:
    # evaluate the reference rank by rank on the SAME shards and average (losses are batch means of equal-size shards).
    ref, rank0 = None, None
    for r in range(world):
        shard = {k: v[r * cfg.batch_size:(r + 1) * cfg.batch_size] for k, v in full.items()}
        o = T.step_gradients(T.Net(cfg, W, torch.float64), 11, shard)
        flat = {m: [t.reshape(-1).numpy() for t in o[k]] for m, k in (("G", "dG"), ("D", "dD"), ("A", "dA"))}
        rank0 = flat if rank0 is None else rank0
        ref = flat if ref is None else {m: [a + b for a, b in zip(ref[m], flat[m])] for m in ref}
    # and, for the G/D tapes (no batch concatenation), directly against ONE process on the concatenated global batch
    og = T.step_gradients(T.Net(cfg_g, W, torch.float64), 5, full)
    store, *_ = _store(cfg)
    from littlegan_amd.eager_trainer import train_weight_range
    a = _args(cfg)
    trained = {m: range(*train_weight_range(a, m, batch_no)) for m in "GDA"}
    assert (len(trained["G"]), len(trained["D"])) == {11: (22, 20), 15: (4, 12), 20: (4, 4)}[batch_no] and len(trained["A"]) == 4
    for m in "GDA":
        for i, ((s, e), t) in enumerate(zip(store.ranges[m], ref[m])):
            if i in trained[m]:   # exchanged: sum over ranks (the 1 / world scale sits in the Adam kernel, dist.py)
                assert np.allclose(got[s:s + t.size], t, rtol=1e-9, atol=1e-13), (m, i)
            else:                 # outside the partition group: untouched by the exchange
                assert np.allclose(got[s:s + t.size], rank0[m][i], rtol=1e-9, atol=1e-13), (m, i)   # (== rank 0's own gradient)
                assert not np.allclose(got[s:s + t.size], t, rtol=1e-3, atol=0) or np.abs(t).max() == 0, (m, i)
    for m, key in (("G", "dG"), ("D", "dD")):
        for i, ((s, e), t) in enumerate(zip(store.ranges[m], og[key])):
            t = t.reshape(-1).numpy()
            if i in trained[m]:
                assert np.allclose(got[s:s + t.size] / world, t, rtol=1e-9, atol=1e-13), (m, np.abs(got[s:s + t.size] / world - t).max())
