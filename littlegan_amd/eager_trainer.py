"""EagerTrainer — host-side mirror of /root/reference/eager_trainer.py on the HIP kernels.

The hot path is `train_step_from_inputs` = the arithmetic of eager_trainer.py:133-168 (the two
GradientTapes replaced by a static backward plan), called by `_train_step` (eager_trainer.py:115-169)
after it has drawn the two batches, the noise and the augmented image.  Data parallelism (new,
BASELINE.json north star): one process per GPU, the three gradient sets are contiguous ranges of one
flat buffer and are all-reduced (RCCL) on a side stream while the next tape's backward runs.
"""
from __future__ import annotations

import json
import os
import time
from typing import Dict, Optional

import torch

from . import ops
from .dist import GradSync
from .model import Adjuster, Discriminator, Generator, ParamStore
from .utils import save_image, soft

ADAM_EPS = 1e-8


def _dist_rank_world():
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return d.get_rank(), d.get_world_size()
    return 0, 1


def _barrier():
    d = torch.distributed
    if d.is_available() and d.is_initialized() and d.get_world_size() > 1:
        d.barrier()


def train_weight_range(args, model: str, batch_no: int):
    """eager_trainer.py:104-113 + part_groups :48-52 as (lo, hi) weight-index ranges."""
    groups = {"G": [(0, 4), (4, 8), (8, 22)], "D": [(0, 12), (12, 16), (16, 20)], "A": [(0, 4)]}[model]
    n_all = {"G": 22, "D": 20, "A": 4}[model]
    if args.use_partition and batch_no % (args.partition_interval + 1) == 0:
        return groups[(batch_no // (args.partition_interval + 1)) % len(groups)]
    return (0, n_all)


class EagerTrainer:
    def __init__(self, args, generator: Generator, discriminator: Discriminator, adjuster: Adjuster, dataset):
        self.args = args
        print(" - Initializing Trainer(Executor)...")
        self.dataset = dataset
        self.adjuster = adjuster
        self.discriminator = discriminator
        self.generator = generator
        self.models = [self.discriminator, self.generator, self.adjuster]
        self.device = generator.device
        if getattr(args, "use_gp", False):  # eager_trainer.py:141-143
            raise NotImplementedError("GP didn't implemented on eager mode")
        self.store = ParamStore(generator, discriminator, adjuster)
        # tf.compat.v1.train.AdamOptimizer x3 (eager_trainer.py:28-30): {beta1_power, beta2_power} per optimizer
        self.opt_cfg = {"G": (args.lr, args.beta_1, args.beta_2), "D": (args.lr, args.beta_1, args.beta_2),
                        "A": (args.lr, 0.9, 0.999)}
        self.opt_state = {m: torch.tensor([b1, b2], dtype=torch.float32, device=self.device)
                          for m, (_, b1, b2) in self.opt_cfg.items()}
        self.losses = {k: torch.zeros(1, dtype=torch.float32, device=self.device) for k in ("gen", "disc", "adj")}
        self.sync = GradSync(self.device)
        self.adam_order = ("D", "G", "A")   # = the launch order of the all-reduces (see train_step_from_inputs)
        self.global_epoch = 1
        self._input_step = 0  # counter window of the device-side step inputs (draw_step_inputs); part of the checkpoint
        self.rank, self.world = _dist_rank_world()
        self._init_dir()
        # eager_trainer.py:36-43: restore the newest checkpoint (weights, the three optimizers' slots and beta powers)
        # and the epoch from status.json when args.restore is set
        if getattr(args, "restore", False) and not getattr(args, "no_io", False) and getattr(args, "result_dir", None):
            latest = self.latest_checkpoint()
            if latest is not None:
                print("Loading Checkpoint...")
                self.load_checkpoint(latest)
                st = os.path.join(args.result_dir, "checkpoint", "status.json")
                if os.path.isfile(st):
                    with open(st) as f:
                        self.global_epoch = json.load(f)["epoch"]
        self._init_test_data()

    # ------------------------------------------------------------------ hot path
    def train_step_from_inputs(self, batch_no: int, inp: Dict[str, torch.Tensor]):
        """inp: real_image_1, real_cond_1, real_image_2, real_cond_2, noise, new_image (device fp32, NHWC).
        Returns (fake_image, adj_image|None, gen_loss, disc_loss, adj_loss|None) — losses are 1-element
        device tensors (no host sync on the hot path)."""
        a = self.args
        G, D, A = self.generator, self.discriminator, self.adjuster
        img1, c1, img2, c2 = inp["real_image_1"], inp["real_cond_1"], inp["real_image_2"], inp["real_cond_2"]
        noise, new_image = inp["noise"], inp["new_image"]
        B = img1.shape[0]
        c = a.cond_dim

        # ---- forward: fake = G(noise, c2); D on the [new_image ; fake] batch (eager_trainer.py:134-137)
        ctx_g: dict = {}
        # D's input batch [new_image ; fake].  A caller that owns the step inputs hands the 2B-image buffer over as
        # inp["disc_input"] with new_image ALREADY in its first half (`_train_step` lets the augmentation kernel write there,
        # bench.py generates its synthetic new_image there): G writes `fake` into the second half and nothing is copied.
        d_in = inp.get("disc_input")
        if d_in is None:
            d_in = torch.empty((2 * B,) + tuple(img1.shape[1:]), dtype=torch.float32, device=self.device)
            d_in[:B].copy_(new_image)
        elif not (tuple(d_in.shape) == (2 * B,) + tuple(img1.shape[1:]) and d_in.dtype == torch.float32 and d_in.is_contiguous()
                  and d_in.device == new_image.device and d_in.data_ptr() == new_image.data_ptr()
                  and tuple(new_image.shape) == tuple(img1.shape) and new_image.is_contiguous()):
            raise ValueError("train_step_from_inputs: inp['disc_input'] must be a contiguous fp32 [2B, H, W, C] buffer whose first B "
                             "images ARE inp['new_image'] (a view of it)")
        fake = G([noise, c2], ctx_g, out=d_in[B:])
        ctx_d: dict = {}
        run_adj = bool(a.train_adj and batch_no > 10)
        # the Adjuster reuses D's encoder maps of `fake` as its skip inputs: keep the fp32 maps only then
        p = D.forward_packed(d_in, ctx_d, keep_maps=run_adj)  # [2B, 1+c]: rows [0,B) real, [B,2B) fake

        # ---- disc tape (eager_trainer.py:139,145): 2*BCE(c1,real_c) + BCE(.98,real_pr) + BCE(.02,fake_pr)
        dz = torch.empty(2 * B, 1 + c, dtype=torch.float32, device=self.device)
        ops.bce_heads_loss(p[:B], c1, soft(1.0), 1.0, 2.0, self.losses["disc"], dz[:B], False)
        ops.bce_heads_loss(p[B:], None, soft(0.0), 1.0, 0.0, self.losses["disc"], dz[B:], True)
        # a partition step differentiates one weight group only (eager_trainer.py:104-113,145): untrained groups get no
        # weight-gradient kernels, no all-reduce, and the backward chain stops where nothing below is asked for
        rng_d = train_weight_range(a, "D", batch_no)
        D.backward(ctx_d, dz, need_wgrad=True, need_input_grad=False, train_range=rng_d)
        self.sync.launch("D", self.store, *self.store.model_range("D", *rng_d))

        # ---- gen tape (eager_trainer.py:140,149): BCE(.98,fake_pr) + BCE(c2,fake_c) + l1*mean|img2-fake|
        dz_g = torch.empty(B, 1 + c, dtype=torch.float32, device=self.device)
        ops.bce_heads_loss(p[B:], c2, soft(1.0), 1.0, 1.0, self.losses["gen"], dz_g, False)
        g_img = D.backward(ctx_d, dz_g, need_wgrad=False, need_input_grad=True, rows=slice(B, 2 * B))
        dpre = torch.empty_like(g_img)
        ops.l1_tanh_loss(img2, fake, g_img, dpre, self.losses["gen"], a.l1_lambda, True)
        rng_g = train_weight_range(a, "G", batch_no)
        G.backward(ctx_g, dpre, train_range=rng_g)
        self.sync.launch("G", self.store, *self.store.model_range("G", *rng_g))

        # ---- adjuster branch (eager_trainer.py:152-164)
        adj_image = None
        if run_adj:
            adj_t_cond, adj_in_cond = ops.adj_conditions(c2, c1)   # concat([c2, c1], 0) and (that + 1) * 0.5, one launch
            ctx_a: dict = {}
            # encoder(fake) was computed by D above with the same weights: hand its 4 maps to the Adjuster
            tails = [m[B:] for m in ctx_d["enc_maps"]]  # fp32 maps (f32 path) / bf16 mirrors + the fp32 top map (bf16 path)
            # Adjuster input = [img1 ; fake]: with the encoder maps of `fake` handed over, only img1 is encoded here
            adj_image = A([img1, adj_in_cond], ctx_a, enc_tails=tails)
            ctx_d2: dict = {}
            p_a = D.forward_packed(adj_image, ctx_d2, keep_maps=False, top_only=True)
            dz_a = torch.empty(2 * B, 1 + c, dtype=torch.float32, device=self.device)
            ops.bce_heads_loss(p_a, adj_t_cond, soft(1.0), 1.0, 1.0, self.losses["adj"], dz_a, False)
            g_adj = D.backward(ctx_d2, dz_a, need_wgrad=False, need_input_grad=True)
            dpre_a = torch.empty_like(g_adj)
            # target = [img2 ; img1]: one call per half instead of a 2B-image concatenation (lambda / 2 over half the elements is
            # the same scale, exactly: the element count is a multiple of a power of two)
            ops.l1_tanh_loss(img2, adj_image[:B], g_adj[:B], dpre_a[:B], self.losses["adj"], 0.5 * a.l1_lambda, True)
            ops.l1_tanh_loss(img1, adj_image[B:], g_adj[B:], dpre_a[B:], self.losses["adj"], 0.5 * a.l1_lambda, True)
            A.backward_own(ctx_a, dpre_a)
            self.sync.launch("A", self.store, *self.store.model_range("A"))

        # ---- optimizer applies.  The reference applies A, D, G (eager_trainer.py:164-168); the three weight ranges, Adam
        # slots and beta-power pairs are disjoint, so any order gives the same bits (tests/test_step_gpu.py::
        # test_adam_order_is_immaterial, tests/test_dp_cpu.py).  Here: the order the all-reduces were launched in, each set
        # waited for on its own right before its Adam — D and G are applied while A's all-reduce (launched last) is on the
        # wire.  D-grads are clipped after the all-reduce, inside the Adam kernel.
        gscale = 1.0 / self.sync.world_size
        for m in self.adam_order:
            if m == "A" and not run_adj:
                continue
            self.sync.wait(m)
            lo, hi = train_weight_range(a, m, batch_no)
            s, e = self.store.model_range(m, lo, hi)
            lr, b1, b2 = self.opt_cfg[m]
            clip = a.clip_range if (m == "D" and a.use_clip) else 0.0
            st = self.store
            ops.clip_adam_update(st.flat[s:e], st.grad[s:e], st.m[s:e], st.v[s:e], self.opt_state[m], lr, b1, b2,
                                 ADAM_EPS, clip, gscale)
            ops.adam_advance(self.opt_state[m], b1, b2)
        self.sync.wait_all()
        self.store.bump()
        return (fake, adj_image, self.losses["gen"], self.losses["disc"], self.losses["adj"] if run_adj else None)

    # ------------------------------------------------------------------ the same step as a replayed HIP graph
    def step_kind(self, batch_no: int):
        """Steps with the same kind enqueue the same kernels on the same shapes: (partition group or -1, Adjuster on)."""
        a = self.args
        part = (batch_no // (a.partition_interval + 1)) % 3 if (a.use_partition and batch_no % (a.partition_interval + 1) == 0) else -1
        return part, bool(a.train_adj and batch_no > 10)

    def graph_step(self, batch_no: int, inp: Dict[str, torch.Tensor]):
        """train_step_from_inputs through a captured HIP graph, one graph per step kind (full step, the three partition
        groups, with / without the Adjuster branch).  The C ABI allocates nothing and never synchronises, so the whole
        step (~330 launches) is capturable; the first step of a kind runs eagerly (it is a real training step and sizes
        every workspace), the graph is captured right after it and every later step of that kind is one replay.
        Results are bit-identical to the eager path (tests/test_step_gpu.py::test_graph_replay_is_bit_exact).
        Single-GPU only: with data parallelism the all-reduces stay on the eager path."""
        if self.sync.enabled:
            if not getattr(self, "_graph_dp_warned", False):   # (round 5: said once instead of silently)
                import warnings
                warnings.warn("graph_step: data parallelism is on — the step runs on the eager path (all-reduces are not captured)",
                              RuntimeWarning, stacklevel=2)
                self._graph_dp_warned = True
            return self.train_step_from_inputs(batch_no, inp)
        kind = self.step_kind(batch_no)
        if not hasattr(self, "_graphs"):
            self._graphs, self._graph_pool, self._graph_seen = {}, None, set()
            self._graph_in = {k: torch.empty_like(v) for k, v in inp.items()}
            if "disc_input" in inp:   # new_image must stay the first half of the static 2B-image buffer
                self._graph_in["new_image"] = self._graph_in["disc_input"][:inp["new_image"].shape[0]]
        for k, v in inp.items():
            if k != "disc_input":     # (its first half arrives as new_image, its second half is written by the step)
                self._graph_in[k].copy_(v)
        if kind not in self._graphs:
            if kind not in self._graph_seen:  # first step of this kind: eager (sizes workspaces, builds every lazy static)
                self._graph_seen.add(kind)
                return self.train_step_from_inputs(batch_no, self._graph_in)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            # the graph records the raw addresses of ops.workspace() buffers: from here on an outgrown buffer is retired, not freed
            ops.pin_workspaces()
            prof_was, ops.Profile.enabled = ops.Profile.enabled, False   # HIP events cannot sit inside a captured graph
            try:
                with torch.cuda.graph(g, pool=self._graph_pool):   # records the launches, executes nothing
                    out = self.train_step_from_inputs(batch_no, self._graph_in)
            finally:
                ops.Profile.enabled = prof_was
            self._graphs[kind] = (g, out)
        g, out = self._graphs[kind]
        g.replay()
        return out

    # ------------------------------------------------------------------ eager_trainer.py:115-169
    def _train_step(self, batch_no, iterator):
        try:
            real_image_1, real_cond_1 = iterator.get_next()
            real_image_2, real_cond_2 = iterator.get_next()
        except StopIteration:  # tf.errors.OutOfRangeError
            return None,
        if not real_cond_1.shape[0] == real_cond_2.shape[0] == self.args.batch_size:
            return False,
        # the augmented batch is written straight into the first half of D's [new_image ; fake] input
        d_in = torch.empty((2 * real_image_1.shape[0],) + tuple(real_image_1.shape[1:]), dtype=torch.float32, device=self.device)
        noise, new_image = self.draw_step_inputs(real_image_1, out=d_in[:real_image_1.shape[0]])
        inp = dict(real_image_1=real_image_1, real_cond_1=real_cond_1, real_image_2=real_image_2,
                   real_cond_2=real_cond_2, noise=noise, new_image=new_image, disc_input=d_in)
        fake, adj, lg, ld, la = self.train_step_from_inputs(batch_no, inp)
        return True, fake, adj, lg, ld, la

    def draw_step_inputs(self, real_image_1, out=None):
        """eager_trainer.py:125-131 on the device: noise ~ N(0,1) and the augmented copy of the first real batch.  All
        draws are counter-based (Philox4x32-10 keyed by args.seed and the rank; one 2^40-block counter window per step),
        so a step's inputs can be regenerated from (seed, rank, step) alone."""
        a = self.args
        self._input_step += 1
        step = self._input_step
        seed = (int(getattr(a, "seed", 0)) << 20) ^ self.rank
        base = step << 40
        noise = ops.randn((a.batch_size, a.noise_dim), seed, base, device=self.device)
        # the scalar draws of the TF ops (one per batch) and the per-image flips come from the block window at 2^39 and are
        # made ON THE DEVICE (lg_augment_drawn): random_brightness(0.02), random_contrast(0.75, 1.003), random_hue(0.03),
        # random_flip_left_right, + 0.1 * N(0, 0.2) from the window at 2^38 — no host synchronisation in the step
        new_image = ops.augment_drawn(real_image_1.contiguous(), 0.02, 0.75, 1.003, 0.03, 0.1 * 0.2, seed,
                                      base + (1 << 39), base + (1 << 38), out=out)
        return noise, new_image

    # ------------------------------------------------------------------ eager_trainer.py:180-229
    def _interrupted(self, signum, f_name):
        """eager_trainer.py:171-178: SIGINT saves an "interrupt" checkpoint + status.json and exits 1."""
        self.save_checkpoint("interrupt")
        with open(os.path.join(self.args.result_dir, "checkpoint", "status.json"), "w") as f:
            json.dump({"epoch": self.global_epoch}, f)
        print("\n Checkpoint has been saved")
        print(signum, f_name)
        import sys
        sys.exit(1)

    def train(self):
        a = self.args
        # data parallel (one process per GPU): every file of the run is written by rank 0 alone — two ranks writing the
        # same checkpoint / image / JSON paths race (and a rank that dies there leaves the others in an all-reduce)
        io = not getattr(a, "no_io", False) and getattr(a, "result_dir", None) and self.rank == 0
        if io:
            import signal
            signal.signal(signal.SIGINT, self._interrupted)
        for e in range(self.global_epoch, a.epoch + 1):
            print("Experiment:", a.exp_name, "Epoch:", e, "Starting...")
            self.global_epoch = e
            iterator = self.dataset.get_new_iterator()
            start_time = time.time()
            seen = 0
            for b in range(1, self.dataset.batches + 1):
                result = self._train_step(b, iterator)
                if result[0] is None:
                    break
                elif not result[0]:
                    continue
                seen += a.batch_size * 2
                if b % a.freq_gen == 0 and self.rank == 0:
                    if io:
                        save_image(result[1], os.path.join(a.result_dir, "train", "gen", "%d-%d.jpg" % (e, b)))
                        if result[2] is not None:
                            save_image(result[2], os.path.join(a.result_dir, "train", "adj", "%d-%d.jpg" % (e, b)))
                    lg, ld = float(result[3]), float(result[4])
                    la = float(result[5]) if result[5] is not None else float("nan")
                    print(f"  [{seen}] LossG {lg:.4f} LossD {ld:.4f} LossA {la:.4f}")
                if b % a.freq_test == 0 and io and self.test_cond is not None:
                    self.predict(self.test_noise, self.test_cond, self.test_image,
                                 os.path.join(a.result_dir, "test", "gen", "%d-%d.jpg" % (e, b)),
                                 os.path.join(a.result_dir, "test", "disc", "%d-%d.json" % (e, b)),
                                 os.path.join(a.result_dir, "test", "adj", "%d-%d.jpg" % (e, b)))
            torch.cuda.synchronize()
            print("Time usage:", time.time() - start_time, "s")
            if io:  # eager_trainer.py:229
                self.save_checkpoint(str(e))
            _barrier()  # the other ranks do not run ahead into the next epoch while rank 0 writes

    # ------------------------------------------------------------------ eager_trainer.py:265-298
    def predict(self, noise, cond, image, gen_image_save_path=None, json_save_path=None, adj_image_save_path=None):
        start_time = time.time()
        gen_image = self.generator([noise, cond])
        torch.cuda.synchronize()
        print("Generate Time", time.time() - start_time, "s")
        if gen_image_save_path is not None:
            save_image(gen_image, gen_image_save_path)
        save = dict()
        save["real_cond"] = cond
        save["real_pr"], save["real_c"] = self.discriminator(image)
        save["fake_pr"], save["fake_c"] = self.discriminator(gen_image)
        mse = lambda t, p: float(((t - p) ** 2).mean(dim=-1).mean(dim=0))
        save["real_pr_mse"] = mse(soft(1.0), save["real_pr"])
        save["real_c_mse"] = mse(cond, save["real_c"])
        save["fake_pr_mse"] = mse(soft(0.0), save["fake_pr"])
        save["fake_c_mse"] = mse(cond, save["fake_c"])
        for x in ["real_cond", "real_pr", "real_c", "fake_c", "fake_pr"]:
            save[x] = torch.round(save[x] * 100).to(torch.int64).cpu().tolist()
        if json_save_path is not None:
            with open(json_save_path, "w") as f:
                json.dump(save, f)
        adj_fake_image, adj_real_image = None, None
        if self.args.train_adj:
            adj_real_image = self.adjuster([image, cond])
            adj_fake_image = self.adjuster([gen_image, cond])
            if adj_image_save_path is not None:
                save_image(torch.cat([adj_real_image, adj_fake_image], 0), adj_image_save_path)
        return gen_image, save, adj_real_image, adj_fake_image

    # ------------------------------------------------------------------ shell (own format; TF checkpoints are out of scope)
    def _init_dir(self):
        rd = getattr(self.args, "result_dir", None)
        if not rd or getattr(self.args, "no_io", False):
            return
        for item in [".", "train/gen", "train/adj", "test/adj", "test/gen", "test/disc", "checkpoint", "log", "sample",
                     "evaluate/gen", "evaluate/adj", "evaluate/disc", "model"]:
            os.makedirs(os.path.join(rd, item), exist_ok=True)

    def _init_test_data(self):
        """eager_trainer.py:65-83: the FIXED evaluation batch {n: noise, c: cond, i: image} of `predict`.  Loaded from
        <test_data_dir>/test_data_<env>.npz when that file exists and args.reuse is set; otherwise drawn once (first
        batch of a fresh iterator + N(0,1) noise) and written there, so every later run evaluates the same batch."""
        import numpy as np
        self.test_noise = self.test_cond = self.test_image = None
        a = self.args
        tdir = getattr(a, "test_data_dir", None)
        npz = os.path.join(tdir, "test_data_" + str(getattr(a, "env", "default")) + ".npz") if tdir else None

        def usable(path):
            """a complete file whose shapes the models accept (the reference loads whatever batch count the file holds and regenerates
            only when G / D / A reject its shapes, :69-83): noise_dim, cond_dim, H, W, C must fit; n, c, i share one leading
            dimension, which need NOT be args.batch_size (a file written under another batch size, or from a short first batch)."""
            try:
                with np.load(path) as d:
                    n, c, i = d["n"], d["c"], d["i"]
            except Exception as e:
                print(f"test data {path}: unreadable ({e}); regenerating")
                return False
            H = a.init_dim * 16
            why = None
            if n.ndim != 2 or n.shape[1] != a.noise_dim:
                why = f"noise {n.shape} does not fit noise_dim {a.noise_dim}"
            elif c.ndim != 2 or c.shape[1] != a.cond_dim:
                why = f"cond {c.shape} does not fit cond_dim {a.cond_dim}"
            elif i.ndim != 4 or tuple(i.shape[1:]) != (H, H, a.image_channel):
                why = f"image {i.shape} does not fit {H}x{H}x{a.image_channel}"
            elif not (n.shape[0] == c.shape[0] == i.shape[0] and n.shape[0] > 0):
                why = f"n, c, i hold {n.shape[0]}, {c.shape[0]}, {i.shape[0]} samples"
            if why:
                print(f"test data {path}: {why}; regenerating")
            return why is None

        # data parallel: ONE decision for all ranks (rank 0 looks, the others are told) — a rank that finds rank 0's half-
        # written file, or loads while others generate, would leave the dataset generators of the ranks one draw apart
        load = bool(npz and getattr(a, "reuse", False) and os.path.isfile(npz) and usable(npz)) if self.rank == 0 else False
        if self.world > 1:
            flag = [load]
            torch.distributed.broadcast_object_list(flag, src=0)
            load = flag[0]
        if load:
            data = np.load(npz)
            to = lambda v: torch.tensor(np.asarray(v, np.float32), device=self.device)
            self.test_noise, self.test_cond, self.test_image = to(data["n"]), to(data["c"]), to(data["i"])
            if self.dataset is not None and self.world > 1:
                # data parallel only: the generating path draws one epoch order — keep every rank's (and every path's) dataset
                # generator in step.  A single process does not draw (the reference does not on reuse, :69-76).
                self.dataset.get_new_iterator()
            return
        if self.dataset is None:
            if getattr(a, "reuse", False) and npz:   # dataset-less modes (condition-sample, export-model): say so, they do not evaluate
                print(f"No fixed evaluation batch: {npz} is missing or does not fit this configuration and there is no dataset to draw "
                      "one from (test_noise / test_cond / test_image stay None)")
            return
        print("No reuse test data, generating...")
        it = self.dataset.get_new_iterator()
        self.test_image, self.test_cond = it.get_next()
        # tf.random.normal([B, noise_dim]) (:82) from the counter-based generator: window 0 is never used by a step
        seed = (int(getattr(a, "seed", 0)) << 20) ^ 0xFFFFF
        self.test_noise = ops.randn((self.test_cond.shape[0], a.noise_dim), seed, 0, device=self.device)
        if npz and self.rank == 0 and not getattr(a, "no_io", False):
            os.makedirs(tdir, exist_ok=True)
            tmp = npz + ".tmp.npz"   # written whole, then renamed: no reader ever sees a partial file
            np.savez_compressed(tmp, n=self.test_noise.cpu().numpy(), c=self.test_cond.cpu().numpy(),
                                i=self.test_image.cpu().numpy())
            os.replace(tmp, npz)
        _barrier()

    # ---- checkpoints (own format: the TF checkpoint format is out of scope, the CONTENT is the reference's:
    # tf.train.Checkpoint(discriminator, generator, adjuster, three optimizers) eager_trainer.py:31-35)
    def checkpoint_state(self) -> dict:
        st = self.store
        return {"format": "littlegan_amd-ckpt-1",
                "names": {m: st.names(m) for m in "GDA"},
                "flat": st.flat.detach().cpu(), "adam_m": st.m.detach().cpu(), "adam_v": st.v.detach().cpu(),
                "beta_powers": {m: t.detach().cpu() for m, t in self.opt_state.items()},
                "epoch": self.global_epoch, "input_step": int(self._input_step)}

    def save_checkpoint(self, tag: str) -> str:
        """Rank 0 writes (weights are identical on every rank after the all-reduced step); other ranks return the path."""
        d = os.path.join(self.args.result_dir, "checkpoint")
        if self.rank != 0:
            return os.path.join(d, f"ckpt-{tag}.pt")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"ckpt-{tag}.pt")
        tmp = path + ".tmp"
        torch.save(self.checkpoint_state(), tmp)
        os.replace(tmp, path)  # a killed save never leaves a truncated "latest"
        with open(os.path.join(d, "checkpoint"), "w") as f:  # same role as TF's "checkpoint" index file
            json.dump({"latest": os.path.basename(path)}, f)
        return path

    def latest_checkpoint(self) -> Optional[str]:
        d = os.path.join(self.args.result_dir, "checkpoint")
        idx = os.path.join(d, "checkpoint")
        if not os.path.isfile(idx):
            return None
        with open(idx) as f:
            p = os.path.join(d, json.load(f)["latest"])
        return p if os.path.isfile(p) else None

    def load_checkpoint(self, path: str):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        st = self.store
        if ck.get("format") != "littlegan_amd-ckpt-1":
            raise ValueError(f"{path}: not a littlegan_amd checkpoint")
        for m in "GDA":
            if ck["names"][m] != st.names(m):
                raise ValueError(f"{path}: weight list of {m} does not match this model configuration")
        if ck["flat"].numel() != st.flat.numel():
            raise ValueError(f"{path}: {ck['flat'].numel()} parameters, this configuration has {st.flat.numel()}")
        st.flat.copy_(ck["flat"])
        st.m.copy_(ck["adam_m"])
        st.v.copy_(ck["adam_v"])
        for m, t in ck["beta_powers"].items():
            self.opt_state[m].copy_(t)
        self._input_step = int(ck.get("input_step", 0))  # a resumed run continues the Philox input stream
        st.bump()

    def export_model_checkpoint(self):
        path = os.path.join(self.args.result_dir, "model", "model.pt")
        torch.save({"names": {m: self.store.names(m) for m in "GDA"}, "flat": self.store.flat.cpu()}, path)
        return path

    def plot(self):
        raise NotImplementedError("plot (keras plot_model / pydot) is UI tooling outside the hot path (SURVEY.md §2 row 5)")
