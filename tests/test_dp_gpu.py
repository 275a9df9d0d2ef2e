"""Data-parallel step with the REAL kernels: two processes share the one GPU of the test box (gloo backend, which
also moves CUDA tensors; RCCL refuses two ranks on one device) and each runs `EagerTrainer.train_step_from_inputs`
on its own half of a global batch, gradients exchanged by littlegan_amd.dist.GradSync on the side stream.
Acceptance: the N-rank result equals the 1-rank step on the concatenated batch within fp32
reduction-order tolerance — every op is per-sample, losses are means of equal-size shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # the spawned ranks import this module by name too
from oracle import np_oracle as O  # noqa: E402
from test_step_gpu import build, dev_inputs, f32_round, perturbed  # noqa: E402

pytestmark = pytest.mark.gpu

CFG = dict(init_dim=2, conv_filter=(64, 32, 32, 32, 32), cond_dim=5, noise_dim=11, batch_size=2)
STEPS = (10, 11)  # a partition step without the Adjuster branch, then a full step with it


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(cfg, world, b):
    return f32_round(O.make_inputs(cfg, cfg.batch_size * world, seed=300 + b))


def _worker(rank, world, port, mfma, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = O.Cfg(**CFG)
        tr = build(cfg, perturbed(cfg, 21), mfma)
        assert tr.sync.enabled and tr.sync.world_size == world
        B = cfg.batch_size
        for b in STEPS:
            full = _inputs(cfg, world, b)
            shard = {k: v[rank * B:(rank + 1) * B] for k, v in full.items()}
            tr.train_step_from_inputs(b, dev_inputs(shard))
        torch.cuda.synchronize()
        np.save(os.path.join(outdir, f"flat_{rank}.npy"), tr.store.flat.cpu().numpy())
        np.save(os.path.join(outdir, f"grad_{rank}.npy"), tr.store.grad.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mfma", ["f32", "bf16"])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path, mfma):
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mfma, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    flat = [np.load(tmp_path / f"flat_{r}.npy") for r in range(world)]
    grad = [np.load(tmp_path / f"grad_{r}.npy") for r in range(world)]
    # every rank ends with identical weights and identical (all-reduced) gradient buffers
    assert np.array_equal(flat[0], flat[1]) and np.array_equal(grad[0], grad[1])

    # one process, global batch: rows ordered like the shards, and the Adjuster pairs samples per shard, so feed the
    # single process rank-major inputs whose Adjuster concat [img1; fake] covers the same samples
    cfg = O.Cfg(**CFG)
    cfg_g = O.Cfg(**{**CFG, "batch_size": cfg.batch_size * world})
    tr = build(cfg_g, perturbed(cfg, 21), mfma)
    for b in STEPS:
        tr.train_step_from_inputs(b, dev_inputs(_inputs(cfg, world, b)))
    torch.cuda.synchronize()
    g1 = tr.store.grad.cpu().numpy()
    w1 = tr.store.flat.cpu().numpy()
    # last step's gradients: DP buffer holds the SUM over ranks of per-rank batch means
    gd = grad[0] / world
    tol = 2e-5 if mfma == "f32" else 2e-3
    for m in "GDA":
        s, e = tr.store.model_range(m)
        num = np.sqrt(np.mean((gd[s:e] - g1[s:e]) ** 2))
        den = np.sqrt(np.mean(g1[s:e] ** 2)) + 1e-30
        assert num / den < tol, (m, num / den)
    # weights after two Adam steps (lr 5e-5): Adam normalises the gradient, so parameters whose true gradient is zero
    # (conv biases in front of an InstanceNormalization) move by +-lr on rounding noise alone -> compare the mean
    assert np.abs(flat[0] - w1).mean() < 0.05 * cfg.lr * len(STEPS)
