#!/usr/bin/env python3
"""bench.py — LittleGAN training-step throughput on MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W [--workload c3|c2] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path (eager_trainer.py:133-168 of the reference: G fwd, D fwd on
[real;fake], disc tape, gen tape, Adjuster branch, D-clip, three Adam applies) on one synthetic batch
already resident in HBM.  Metric (BASELINE.json): 128x128 images/sec, images = batch_size per step
(one batch of fakes generated per step; the reference's Progbar counts 2*batch_size consumed samples).
Workloads:  c3 (default; the config the metric is quoted on): 128^2, B=256/GPU, bf16 MFMA, G+D+Adj
            c2: 128^2, B=64/GPU, exact-f32 MFMA, G+D only (train_adj off)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK = {"bf16": 2500.0, "f32": 157.3}  # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md:41-43
# algorithmic conv/convT/dense FLOPs per image of the per-device batch at 128^2, c=40
GFLOP_PER_IMAGE = {"c3": 27.03, "c2": 13.25, "c5": 108.1}  # c5: 256^2


def make_args(workload, device, dtype=None):
    c3 = workload in ("c3", "c5")  # c5 = the C3 step at 256x256 (init_dim 16): the per-GPU share of BASELINE configs[4]
    return SimpleNamespace(
        batch_size=256 if c3 else 64, image_channel=3, noise_dim=93, init_dim=16 if workload == "c5" else 8,
        conv_filter=[384, 256, 128, 64, 32],
        kernel_size=5, leaky_alpha=0.3, dropout_rate=0.5, l1_lambda=0.02, lr=5e-5, beta_1=0.5, beta_2=0.9,
        use_gp=False, use_clip=True, clip_range=0.5, use_partition=True, partition_interval=4,
        train_adj=c3, cond_dim=40, mfma_dtype=dtype or ("bf16" if c3 else "f32"), device=device, seed=0, no_io=True)


def synthetic_inputs(args, device, rank):
    """images U(-1,1), conds soft(+-1), noise N(0,1); seed 1234 + rank."""
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    B, H = args.batch_size, args.init_dim * 16
    img = lambda: (torch.rand(B, H, H, 3, generator=g) * 2 - 1).to(device)
    cond = lambda: (0.96 * (2.0 * torch.randint(0, 2, (B, args.cond_dim), generator=g).float() - 1.0) + 0.02).to(device)
    out = dict(real_image_1=img(), real_cond_1=cond(), real_image_2=img(), real_cond_2=cond(),
               noise=torch.randn(B, args.noise_dim, generator=g).to(device))
    # new_image is resident as the first half of the discriminator's [new_image ; fake] input batch — where the training loop's
    # augmentation kernel writes it (EagerTrainer._train_step); the Generator writes `fake` into the second half every step
    d_in = torch.empty(2 * B, H, H, 3, dtype=torch.float32, device=device)
    d_in[:B].copy_(img())
    out["disc_input"], out["new_image"] = d_in, d_in[:B]
    return out


def _cpu_time_step(cfg_kw, Bc, min_steps, budget_s, max_steps=20):
    """Times the oracle's torch-CPU fp32 restatement of the step on the host cores: (images/sec, steps, s/step)."""
    from oracle import np_oracle as O
    from oracle import torch_oracle as T
    cfg = O.Cfg(batch_size=Bc, cond_dim=40, **cfg_kw)
    W = O.init_weights(cfg, 0)
    tr = T.Trainer(cfg, W, dtype=torch.float32)
    inp = {k: torch.tensor(v, dtype=torch.float32) for k, v in O.make_inputs(cfg, Bc, 1234).items()}
    for w in range(3):  # >= 3 warm-up steps (BASELINE.md par. 3): allocations, oneDNN primitives
        tr.step(11 + w, inp)
    n, t0 = 0, time.perf_counter()
    while n < min_steps or (time.perf_counter() - t0 < budget_s and n < max_steps):
        tr.step(14 + n, inp)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return Bc / dt, n, dt


CPU_CONFIGURED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "cpu_baseline_configured.json")


def cpu_baseline(workload, full=False):
    """`value` = the oracle's torch-CPU fp32 restatement of the SAME step at the CONFIGURED batch, measured once on a GPU box's host cores
    by scripts/cpu_baseline_full.py and cached in profiles/cpu_baseline_configured.json (C3: 48 s per step — beyond the bounded budget
    of a default run); `live_sample` = the same restatement timed NOW on this box on a bounded sample (B = 8, ~10 s) — the figure that
    says whether this box's host is comparable; plus the two CPU-sized configurations SURVEY.md par. 8d names (C1: 64x64, B=16, full
    step; C2: 128x128, G+D only — at B=16 by default, at the full B=64 with --cpu-baseline-full).  >= 3 warm-up steps each.
    Label: CPU restatement, not TensorFlow (TF 1.15 cannot run in this pipeline)."""
    Bc = 8
    idim = 16 if workload == "c5" else 8
    v, n, dt = _cpu_time_step(dict(train_adj=(workload != "c2"), init_dim=idim), Bc, 10, 10.0)
    live = {"value": round(v, 3), "unit": "images/sec", "cores": torch.get_num_threads(),
            "sample": f"torch-CPU fp32 restatement (oracle/torch_oracle.py) of the same {workload} step at {idim * 16}x{idim * 16}, "
                      f"batch {Bc}, {n} timed steps after 3 warm-up, {dt:.2f} s/step, nproc={os.cpu_count()}"}
    key = {"c3": "C3 (128x128, B=256, G+D+Adj)", "c2": "C2 (128x128, B=64, G+D)"}.get(workload)
    cached = None
    if key and os.path.exists(CPU_CONFIGURED):
        with open(CPU_CONFIGURED) as f:
            cj = json.load(f)
        if key in cj:
            cached = (cj[key], cj.get("threads"), cj.get("nproc"))
    if cached:
        c, thr, npr = cached
        out = {"value": c["images_per_sec"], "unit": "images/sec", "cores": thr, "kind": "port",
               "sample": f"torch-CPU fp32 restatement (oracle/torch_oracle.py) of the {workload} step at its CONFIGURED batch: {key}, {c['timed_steps']} timed steps after "
                         f"{c['warmup_steps']} warm-up, {c['s_per_step']} s/step on {thr} threads (nproc={npr}) of a GPU box — cached in profiles/cpu_baseline_configured.json "
                         "(scripts/cpu_baseline_full.py; too long for the bounded budget of a default run); live_sample is the same restatement timed in THIS run at batch 8",
               "live_sample": live}
    else:   # no configured-batch figure for this workload: the bounded live sample is the value
        out = dict(live, kind="port")
    v1, n1, dt1 = _cpu_time_step(dict(train_adj=True, init_dim=4), 16, 10, 5.0)
    b2 = 64 if full else 16
    v2, n2, dt2 = _cpu_time_step(dict(train_adj=False, init_dim=8), b2, 10, 5.0, max_steps=10)
    out["other_shapes"] = {
        "C1 (64x64, B=16, cond 40, G+D+Adj)": {"images_per_sec": round(v1, 3), "steps": n1, "s_per_step": round(dt1, 3)},
        f"C2 (128x128, B={b2}, f32, G+D only)": {"images_per_sec": round(v2, 3), "steps": n2, "s_per_step": round(dt2, 3)}}
    return out


def precision_note(workload, dtype):
    """What the line's dtype means for the results, and the other side of the trade (measured figures from the newest committed
    collection of the other dtype, if there is one)."""
    import glob, json, os, re
    here = os.path.dirname(os.path.abspath(__file__))
    other = "f32" if dtype == "bf16" else "bf16"
    name = f"bench_{workload}_f32.json" if other == "f32" else f"bench_{workload}.json"
    best = None
    for p in glob.glob(os.path.join(here, "profiles", f"r*_{name}")):
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(p))
        if m and (best is None or (int(m.group(1)), m.group(2)) > best[0]):
            best = ((int(m.group(1)), m.group(2)), p)
    ref = None
    if best:
        try:
            d = json.load(open(best[1]))
            if d.get("dtype") == other:   # (c2's plain file IS the f32 run: nothing to quote then)
                ref = {"dtype": other, "ms_per_step": d["ms_per_step"], "value": d["value"], "source": os.path.relpath(best[1], here)}
        except Exception:
            ref = None
    if dtype == "bf16":
        what = ("bf16 MFMA operands, fp32 accumulation: losses within 2e-2, images within 2.4e-2 of the fp64 restatement "
                "(tests/test_step_gpu.py); the north star's 1e-4 is met by the exact-f32 step only (--dtype f32; 2e-5 measured)")
    else:
        what = "exact fp32 MFMA (v_mfma_f32_32x32x2_f32): losses within 2e-5 of the fp64 restatement (tests/test_launch_shapes_gpu.py)"
    return {"this_line": what, "other_dtype_same_workload": ref}


def traffic_files():
    """profiles/r<N>[suffix]_pmc_traffic.json, oldest first: r5_..., then r5b_... (a second collection of a round) — never a parse error on a name."""
    import glob
    import re

    def key(f):
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(f))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")), key=key)


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N FRESH rank processes (torch.distributed.run, one per GPU)
    before this process has touched the GPU, relay their output (rank 0 prints the JSON line) and exit with the
    launcher's code — a failed rank is a failed bench, never a re-exec of a process that holds the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    # own process group: on expiry of the wall limit (a rank stuck in init_process_group or in a collective) the launcher
    # AND its rank processes are ended by group id — never by pattern — and the bench fails; nothing is started again
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)

    def stop():
        import signal
        for sig, grace in ((signal.SIGTERM, 20), (signal.SIGKILL, 10)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                return
            try:
                proc.wait(timeout=grace)
                return
            except subprocess.TimeoutExpired:
                continue
    try:
        rc = proc.wait(timeout=a.wall_limit)
    except subprocess.TimeoutExpired:
        stop()
        print(f"bench.py: the {a.gpus}-rank run exceeded --wall-limit {a.wall_limit} s and was terminated", file=sys.stderr)
        raise SystemExit(124)
    except BaseException:
        stop()
        raise
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # >= 50 timed steps after 15 warm-up (partition steps included)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--workload", choices=["c3", "c2", "c5"], default="c3")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default=None,
                    help="MFMA operand type (default: bf16 for c3 / c5, f32 for c2).  `--workload c3 --dtype f32` = the G+D+Adj "
                         "B=256 step at the north star's 1e-4 loss tolerance (exact-f32 products)")
    ap.add_argument("--dp-contention", default=None,
                    help="one-GPU rehearsal of the all-reduces' CU footprint: comma list of workgroup counts K (e.g. 8,16,32,64); "
                         "at the three GradSync.launch points K streaming read-add-write blocks run on the side stream")
    ap.add_argument("--reserve-cus", type=int, default=None, help="CUs the persistent kernels leave free (default: 0 at N=1, dist.RESERVED_CUS_DP at N>1)")
    ap.add_argument("--wall-limit", type=int, default=900, help="seconds the self-spawned N-rank run may take before it is terminated")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="time the C2 CPU shape at its full batch 64 (minutes)")
    ap.add_argument("--no-graph-leg", action="store_true", help="skip the extra hipGraph-replay timing (profilers that collect counters cannot follow a graph capture)")
    ap.add_argument("--graph", action="store_true", help="time EagerTrainer.graph_step (captured HIP graphs) instead of eager launches")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo: rehearsal of the N>1 path")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo; RCCL refuses it)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)  # does not return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if a.share_gpu:
        if a.backend != "gloo":
            raise SystemExit("--share-gpu needs --backend gloo (RCCL does not run two ranks on one device)")
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("LG_DIST_TIMEOUT_S", "300")))   # a missing rank fails the bench instead of hanging it
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)

    from littlegan_amd import ops
    from littlegan_amd.eager_trainer import EagerTrainer
    from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator
    args = make_args(a.workload, str(device), a.dtype)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # stdout carries exactly ONE JSON line
        decoder, encoder = Decoder(args), Encoder(args)
        gen = Generator(args, decoder)
        disc = Discriminator(args, encoder)
        adj = Adjuster(args, disc, gen)
        tr = EagerTrainer(args, gen, disc, adj, None)
    if world > 1:  # identical initial weights on every rank
        import torch.distributed as dist
        dist.broadcast(tr.store.flat, src=0)
        tr.store.bump()
    inp = synthetic_inputs(args, device, rank)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if a.reserve_cus is not None:
        tr.sync.reserve_cus(a.reserve_cus)
    b0 = 11  # batch_no > 10 so the Adjuster branch runs (eager_trainer.py:152); every 5th step is a partition step
    step = tr.graph_step if a.graph else tr.train_step_from_inputs
    for i in range(a.warmup):
        step(b0 + i, inp)
    barrier()
    if not a.graph:
        ops.Profile.start()
    tr.sync.time_waits = world > 1
    t0 = time.perf_counter()
    # The roofline's per-launch HIP events (two per conv-class launch, on the launch stream) are recorded in every PROF_EVERY-th step
    # of the timed region: measured in round 4 (scripts/probe/event_cost.py), events around every launch of every step cost the eager
    # step 0.27 - 0.30 ms (2.4 %); one step in four keeps >= 12 of the default 50 steps (both step kinds) at a quarter of that.
    PROF_EVERY = 4
    n_prof_steps = 0
    for i in range(a.steps):
        if not a.graph:
            ops.Profile.enabled = (i % PROF_EVERY == 0)
            n_prof_steps += int(ops.Profile.enabled)
        step(b0 + a.warmup + i, inp)
    if not a.graph:
        ops.Profile.enabled = True
    dt_host = time.perf_counter() - t0  # host time to ENQUEUE the K steps (no synchronisation inside a step): below ms_per_step = the GPU is the bound
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0   # this rank's own time for its K steps (before the closing barrier)
    barrier()
    dt = time.perf_counter() - t0
    tr.sync.time_waits = False          # only the timed region's waits count (the roofline / graph legs below add none)
    n_wait_events = len(tr.sync.wait_events)
    if a.graph:  # per-launch HIP events cannot sit inside a replayed graph: the roofline leg times 5 eager steps afterwards
        ops.Profile.start()
        for i in range(5):
            tr.train_step_from_inputs(b0 + a.warmup + a.steps + i, inp)
        torch.cuda.synchronize()
    prof = ops.Profile.stop()
    losses = {k: float(v.item()) for k, v in tr.losses.items()}  # of the last TIMED step (before the graph-replay leg below)
    graph_ms = None
    profiled = any("rocprof" in (os.environ.get(k) or "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if world == 1 and not a.graph and a.workload == "c3" and args.mfma_dtype == "bf16" and not a.no_graph_leg and not profiled:
        # the same steps replayed from captured HIP graphs (EagerTrainer.graph_step, bit-identical results): reported beside
        # the headline number, which stays the eager one (per-launch HIP events cannot sit inside a replayed graph)
        bg = b0 + a.warmup + a.steps
        for i in range(30):  # every step kind at least twice: first eager, then captured
            tr.graph_step(bg + i, inp)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for i in range(30):
            tr.graph_step(bg + 30 + i, inp)
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - tg) / 30 * 1e3
    # ---- legs AFTER the headline region (they never touch `value`) --------------------------------------------------
    def timed_steps(n, first):
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for i in range(n):
            tr.train_step_from_inputs(first + i, inp)
        torch.cuda.synchronize()
        return (time.perf_counter() - t_) / n * 1e3

    b_next = b0 + a.warmup + a.steps + 100
    clock = None
    if world == 1 and not profiled:
        # Shader clock HELD under this step (after the timed region; d s_memtime / d s_memrealtime x 100 MHz):
        #  (a) IN-KERNEL census: every block of the step's dominant kernel template (conv_down3, all its forms) adds the two counters'
        #      increments over its own life (~100 us) to a buffer during 20 further steps (lg_set_clock_census) — the clock those
        #      blocks ran at inside the real step;
        #  (b) a 20-us sampling wave on the COMPUTE stream right behind every conv-class launch of 5 further steps (lg_clock_sample):
        #      the clock an otherwise idle chip shows at those points — an upper bound, the power controller moves it within micro-
        #      seconds of the load ending.
        from littlegan_amd import _lib as lib
        L_ = lib.load()
        cen = torch.zeros(3, dtype=torch.int64, device=device)
        timed_steps(5, b_next)
        lib.check(L_.lg_set_clock_census(cen.data_ptr()), "lg_set_clock_census")
        try:
            ms_clk = timed_steps(20, b_next + 5)
        finally:
            lib.check(L_.lg_set_clock_census(0), "lg_set_clock_census")
        o = cen.tolist()
        if o[1] > 0:
            clock = {"in_kernel_mhz": round(o[0] / o[1] * 100.0, 1), "blocks": o[2], "kernel": "conv_down3_kernel (all forms)",
                     "leg_ms_per_step": round(ms_clk, 3),
                     "method": "sum d s_memtime / sum d s_memrealtime x 100 MHz over the block lives of the step's dominant kernel template, 20 further steps after the timed region"}
        out3 = torch.zeros(3, dtype=torch.int64, device=device)
        ops.Profile.start()
        ops.Profile.after_conv = lambda: lib.check(L_.lg_clock_sample(out3.data_ptr(), 20, torch.cuda.current_stream().cuda_stream), "lg_clock_sample")
        try:
            ms_smp = timed_steps(5, b_next + 30)
        finally:
            ops.Profile.after_conv = None
            ops.Profile.stop()
        o3 = out3.tolist()
        if o3[1] > 0:
            clock = dict(clock or {}, after_conv_launches={"mean_mhz": round(o3[0] / o3[1] * 100.0, 1), "samples": o3[2], "leg_ms_per_step": round(ms_smp, 3),
                                                             "method": "20-us sampling wave on the compute stream behind every conv-class launch of 5 further steps (upper bound: chip otherwise idle during the sample)"})
        b_next += 100
    contention = None
    ks = [int(k) for k in a.dp_contention.split(",")] if a.dp_contention else ([32] if (world == 1 and a.workload == "c3" and args.mfma_dtype == "bf16" and not profiled) else [])
    if world == 1 and ks:
        # one-GPU rehearsal of data parallelism's CU contention: K streaming read-add-write workgroups on the side
        # stream at the three GradSync.launch points, with the persistent grids at full size and with K CUs left free
        contention = {"baseline_ms_per_step": round(timed_steps(20, b_next), 3), "threads_per_workgroup": 512, "passes": 2, "runs": []}
        was = tr.sync.reserved_cus
        for k in ks:
            for reserve in (0, k):
                tr.sync.reserve_cus(reserve)
                tr.sync.rehearse(k)
                timed_steps(5, b_next)
                tr.sync.wait_events, tr.sync.time_waits = [], True
                ms_k = timed_steps(20, b_next + 5)
                tr.sync.time_waits = False
                stall = sum(e0.elapsed_time(e1) for e0, e1 in tr.sync.wait_events) / 20
                contention["runs"].append({"workgroups": k, "reserved_cus": reserve, "ms_per_step": round(ms_k, 3),
                                           "wait_stall_ms_per_step": round(stall, 3)})
        tr.sync.rehearse(0)
        tr.sync.reserve_cus(was)
    dp_info = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank view beside the MAX-reduced figure: each rank's own ms/step and how long its compute stream stalled in
        # GradSync.wait_all() (event pair on the compute stream; 0 = the three all-reduces were hidden under the backward)
        wait_ms = sum(e0.elapsed_time(e1) for e0, e1 in tr.sync.wait_events[:n_wait_events]) / max(a.steps, 1)
        mine = torch.tensor([dt_own / a.steps * 1e3, wait_ms], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        own = [float(x[0]) for x in allr]
        waits = [float(x[1]) for x in allr]
        dp_info = {"per_rank_ms_per_step": {"min": round(min(own), 3), "max": round(max(own), 3)},
                   "allreduce_wait_ms_per_step": {"min": round(min(waits), 3), "max": round(max(waits), 3)},
                   "reserved_cus": tr.sync.reserved_cus, "adam_order": "".join(tr.adam_order) + " (each set right after its own wait)",
                   "backend": a.backend, "launch": "eager (graph_step falls back to eager launches under data parallelism)",
                   "allreduce_payload_mb": {m: round((tr.store.model_range(m)[1] - tr.store.model_range(m)[0]) * 4 / 1e6, 2) for m in "DGA"}}

    if rank == 0:
        ms = dt / a.steps * 1e3
        gb = args.batch_size * world
        value = gb / (dt / a.steps)
        dt_name = args.mfma_dtype
        # ops.Profile keys are "<contraction class>:<kernel template the C side launched>" (lg_last_kernel)
        by_kernel, by_class = {}, {}
        for key, (n_, fl_, sec_) in prof.items():
            cls, _, kern = key.partition(":")
            for d, k in ((by_kernel, kern or cls), (by_class, cls)):
                r = d.setdefault(k, [0, 0.0, 0.0])
                r[0] += n_; r[1] += fl_; r[2] += sec_
        # dominant kernel = the kernel template with the largest measured time in the timed region
        tag, (n, fl, sec) = max(by_kernel.items(), key=lambda kv: kv[1][2])
        ach = fl / sec / 1e12
        # HBM-side bytes per launch of that kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) cannot run inside this process;
        # they are collected on this same command and committed (scripts/collect_profiles.sh -> scripts/pmc_traffic.py)
        traffic, traffic_src = None, None
        import glob
        tfiles = traffic_files()
        if a.workload == "c3" and dt_name == "bf16" and world == 1 and tfiles:
            tpath = tfiles[-1]   # the newest round's counter passes
            rec = json.load(open(tpath)).get(tag)
            if rec:
                traffic, traffic_src = rec["bytes_per_launch"], f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, offline; goes stale when the kernel changes)"
        conv_sec = sum(v[2] for v in prof.values())
        conv_fl = sum(v[1] for v in prof.values())
        nsteps_prof = 5 if a.graph else max(n_prof_steps, 1)
        out = {
            "metric": {"c3": "128x128 CelebA-shaped images/sec (G+D+Adj step)", "c2": "128x128 CelebA-shaped images/sec (G+D step)",
                       "c5": "256x256 CelebA-shaped images/sec (G+D+Adj step)"}[a.workload],
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(dt_host / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dt_name, "data": "synthetic",
            "config": {"workload": {"c3": "C3: 128x128x3 synthetic CelebA, batch 256/GPU, bf16 MFMA conv/transposed-conv + Adjuster branch" if dt_name == "bf16" else
                                          "C3 at exact f32 (not a BASELINE config; the north star's 1e-4 loss tolerance): 128x128x3 synthetic CelebA, batch 256/GPU, f32 MFMA + Adjuster branch",
                                    "c2": "C2: 128x128x3 synthetic CelebA, batch 64/GPU, exact-f32 MFMA, G+D step only",
                                    "c5": "C5 (per-GPU share): 256x256x3 synthetic CelebA, batch 256/GPU, bf16 MFMA + Adjuster branch"}[a.workload],
                       "global_batch": gb, "per_gpu_batch": args.batch_size, "image": args.init_dim * 16, "cond_dim": 40,
                       "parallelism": f"dp{world}", "consumed_samples_per_step": 2 * gb,
                       "launch": "hipGraph replay (one graph per step kind)" if a.graph else "eager launches"},
            "roofline": {"bound": "mfma", "kernel": tag, "achieved": round(ach, 2), "peak": PEAK[dt_name], "unit": "TFLOP/s",
                         "frac": round(ach / PEAK[dt_name], 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "launches": n, "avg_launch_ms": round(sec / n * 1e3, 4),
                         "all_kernels": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 2), "frac": round(v[1] / v[2] / 1e12 / PEAK[dt_name], 4),
                                             "avg_launch_ms": round(v[2] / v[0] * 1e3, 4), "ms_per_step": round(v[2] / nsteps_prof * 1e3, 3)}
                                         for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][2])},
                         "all_conv_kernels": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 2),
                                                  "ms_per_step": round(v[2] / nsteps_prof * 1e3, 3)} for k, v in by_class.items()},
                         "event_timed_steps": nsteps_prof, "event_sampling": ("the 5 eager steps after the graph-replayed region" if a.graph else f"every {PROF_EVERY}th step of the timed region"),
                         "conv_share_of_step": round(conv_sec / (dt / a.steps * nsteps_prof), 3),
                         "conv_tflops_overall": round(conv_fl / conv_sec / 1e12, 2),
                         "step_algorithmic_tflops": round(GFLOP_PER_IMAGE[a.workload] * args.batch_size / ms, 2)},
            "graph_replay": None if graph_ms is None else {"ms_per_step": round(graph_ms, 3), "value": round(args.batch_size / graph_ms * 1e3, 2),
                                                               "note": "same steps, one captured hipGraph per step kind, 30 timed steps after the headline region"},
            "data_parallel": dp_info,   # null at one GPU
            # one-GPU rehearsal of the all-reduces' CU footprint: a SYNTHETIC side-stream probe (lg_contention_probe: streaming read-add-write
            # workgroups at the three GradSync.launch points) — NOT RCCL's kernels, channel count or xGMI traffic
            "dp_contention": None if not contention else dict(contention, what="synthetic side-stream probe (lg_contention_probe), not RCCL"),
            "clock": clock,
            "losses_last_step": losses,
            "parity": "checked against the in-repo fp64 restatement (tests/); parity to TensorFlow 1.15 is UNPINNED",
            # the headline's precision and the price of the north star's own tolerance, side by side
            "precision": precision_note(a.workload, args.mfma_dtype),
        }
        if not a.no_cpu_baseline and world == 1:  # reported at N=1 only (rank 0), on a bounded sample
            out["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_baseline_full)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
