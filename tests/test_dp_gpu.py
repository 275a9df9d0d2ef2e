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


def _join_all(procs, timeout):
    """Join every rank; on a timeout or a failure terminate exactly the processes this test started."""
    try:
        for p in procs:
            p.join(timeout=timeout)
            assert p.exitcode == 0, f"rank process exit code {p.exitcode}"
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=20)
                if p.is_alive():
                    p.kill()


def _worker(rank, world, port, mfma, outdir, backend="gloo"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":  # RCCL over xGMI: one rank per GPU
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
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


def _dp_params():
    out = [pytest.param("f32", "gloo"), pytest.param("bf16", "gloo")]
    two = torch.cuda.is_available() and torch.cuda.device_count() >= 2
    out.append(pytest.param("bf16", "nccl", marks=pytest.mark.skipif(not two, reason="RCCL path needs >= 2 GPUs (one rank per device)")))
    return out


@pytest.mark.parametrize("mfma,backend", _dp_params())
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path, mfma, backend):
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mfma, str(tmp_path), backend)) for r in range(world)]
    for p in procs:
        p.start()
    _join_all(procs, 600)
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


def _train_worker(rank, world, port, outdir):
    """train() with file I/O enabled under data parallelism (the advisor's rank-safety case)."""
    import torch.distributed as dist
    from types import SimpleNamespace
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from littlegan_amd.dataset import CelebA
        from littlegan_amd.eager_trainer import EagerTrainer
        from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator
        a = SimpleNamespace(batch_size=2, image_channel=3, noise_dim=5, init_dim=2, conv_filter=[32, 32, 32, 32, 32],
                            kernel_size=5, leaky_alpha=0.3, dropout_rate=0.5, l1_lambda=0.02, lr=5e-5, beta_1=0.5, beta_2=0.9,
                            use_gp=False, use_clip=True, clip_range=0.5, use_partition=True, partition_interval=4,
                            train_adj=True, attr=[1, 2, 3], cond_dim=3, mfma_dtype="f32", device="cuda", seed=3,
                            synthetic=True, synthetic_images=2 * 2 * 2 * 12, image_dim=32, image_path=None, image_ext="jpg",
                            attr_path=None, env="dp", reuse=False, restore=False, no_io=False, exp_name="dp", epoch=2,
                            freq_gen=4, freq_test=6, test_data_dir=os.path.join(outdir, "td"),
                            result_dir=os.path.join(outdir, "res"))
        dec, enc = Decoder(a), Encoder(a)
        g = Generator(a, dec)
        d = Discriminator(a, enc)
        ds = CelebA(a)
        assert ds.batches == 48 // world and ds.world == world
        order = ds.get_new_iterator().order
        np.save(os.path.join(outdir, f"order_{rank}.npy"), np.asarray(order))
        tr = EagerTrainer(a, g, d, Adjuster(a, d, g), ds)
        dist.broadcast(tr.store.flat, src=0)
        tr.store.bump()
        tr.train()
        torch.cuda.synchronize()
        np.save(os.path.join(outdir, f"tflat_{rank}.npy"), tr.store.flat.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_train_loop_with_file_io_is_rank_safe(tmp_path):
    """Two ranks run EagerTrainer.train() on one result directory: rank 0 alone writes checkpoints / images / JSON, the
    ranks read disjoint batches, finish both epochs and hold identical weights."""
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    _join_all(procs, 600)
    f0, f1 = np.load(tmp_path / "tflat_0.npy"), np.load(tmp_path / "tflat_1.npy")
    assert np.array_equal(f0, f1)
    o0, o1 = np.load(tmp_path / "order_0.npy"), np.load(tmp_path / "order_1.npy")
    assert len(o0) == len(o1) == 24 and not set(o0.tolist()) & set(o1.tolist())   # disjoint shards, equal step counts
    ck = tmp_path / "res" / "checkpoint"
    assert (ck / "ckpt-1.pt").is_file() and (ck / "ckpt-2.pt").is_file() and not list(ck.glob("*.tmp"))
    assert (tmp_path / "res" / "train" / "gen" / "2-4.jpg").is_file() and (tmp_path / "res" / "test" / "disc" / "1-6.json").is_file()
    state = torch.load(ck / "ckpt-2.pt", map_location="cpu", weights_only=True)
    assert state["input_step"] == 24   # 12 steps per epoch per rank (24 batches, 2 per step), two epochs: the Philox input stream position is saved


def _rccl_single_worker(outdir):
    """Fresh process (no GPU call before init_process_group): a world-size-1 `nccl` (= RCCL) group on cuda:0."""
    import datetime
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
    try:
        cfg = O.Cfg(**CFG)
        W = perturbed(cfg, 23)
        tr_p, tr_s = build(cfg, W, "bf16"), build(cfg, W, "bf16")
        assert not tr_p.sync.enabled
        tr_s.sync.force_enable()
        assert tr_s.sync.enabled and tr_s.sync.world_size == 1 and tr_s.sync.comm_stream is not None
        for b in (10, 11, 12):   # a partition step, two full steps with the Adjuster branch
            inp = dev_inputs(_inputs(cfg, 1, b))
            rp = tr_p.train_step_from_inputs(b, inp)
            rs = tr_s.train_step_from_inputs(b, inp)
            torch.cuda.synchronize()
            assert not tr_s.sync._pending
            for x, y in zip(rp, rs):
                assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), b
            for x, y in ((tr_p.store.flat, tr_s.store.flat), (tr_p.store.m, tr_s.store.m), (tr_p.store.v, tr_s.store.v), (tr_p.store.grad, tr_s.store.grad)):
                assert torch.equal(x, y), b
        # graph_step under data parallelism: the eager path, and it says so once
        import warnings
        inp = dev_inputs(_inputs(cfg, 1, 13))
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            rs = tr_s.graph_step(13, inp)
            rp = tr_p.train_step_from_inputs(13, inp)
            torch.cuda.synchronize()
            for x, y in zip(rp, rs):
                assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), "graph_step under DP"
            tr_s.graph_step(14, dev_inputs(_inputs(cfg, 1, 14)))
        assert sum("eager path" in str(w.message) for w in caught) == 1, [str(w.message) for w in caught]
        loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libnccl" in ln]
        with open(os.path.join(outdir, "rccl_ok.txt"), "w") as f:
            f.write("\n".join(sorted(set(loaded))) or "none")
    finally:
        dist.destroy_process_group()


def test_rccl_world_size_one_is_bit_identical(tmp_path):
    """De-risks the first multi-GPU run on the one-GPU box: librccl loads, a communicator builds with
    HSA_ENABLE_IPC_MODE_LEGACY=0, three steps run their three all-reduces per step on the side stream with the per-set waits in
    front of each Adam, `destroy_process_group` returns — and every bit of weights, Adam slots, gradients and outputs equals the
    plain step (an all-reduce over one rank is the identity, 1 / world = 1)."""
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_single_worker, args=(str(tmp_path),))
    p.start()
    _join_all([p], 300)
    libs = (tmp_path / "rccl_ok.txt").read_text()
    assert "rccl" in libs or "nccl" in libs, libs   # the collective library really is in the process
