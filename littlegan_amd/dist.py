"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI; "gloo" on CPU for tests).  The reference has no distributed code at all; the step is
pure data parallel because InstanceNormalization has no batch statistics, so the ONLY exchange is the
mean of the three gradient sets.  Each set is one contiguous range of the flat gradient buffer
(model.ParamStore) => one large all-reduce per set (G 27 MB, D 18 MB, A 4 MB at 128^2), launched on a side
stream as soon as its tape's backward has been enqueued, overlapping the next tape's kernels.
The 1/world_size scale and the D-clip are applied afterwards inside the Adam kernel; the compute stream waits
per SET (`wait(name)`) right before that set's Adam — D and G are applied while A's all-reduce is still on the wire.

CU budget: RCCL's ring kernels need CUs at exactly the points `launch` fires, while the persistent conv kernels size their
grids to every CU.  `reserve_cus(n)` shrinks the persistent grids (lg_set_reserved_cus) so that communication workgroups would
find free CUs; `rehearse(K)` is the one-GPU rehearsal of the situation (bench.py --dp-contention): instead of an all-reduce,
`launch` puts K streaming read-add-write workgroups over a range of the same size on the side stream (lg_contention_probe).
MEASURED (round 4, one MI355X, C3 step 11.82 ms): K = 8 / 16 / 32 / 64 side workgroups cost +4.6 / +3.6 / +2.9 / +2.8 %
with the grids at full size and +7.5 / +7.0 / +9.0 / +14.4 % with K CUs left free — a displaced persistent block is cheaper than a
CU that idles for the whole step, so the default reservation under data parallelism is 0 (LG_RESERVED_CUS overrides)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# CUs left to the communication kernels when world > 1: 0 — measured on the one-GPU rehearsal (module docstring);
# LG_RESERVED_CUS overrides.
RESERVED_CUS_DP = 0


class GradSync:
    def __init__(self, device):
        self.device = torch.device(device)
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world_size = dist.get_world_size() if self.enabled else 1
        self.on_gpu = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.enabled and self.on_gpu) else None
        self._pending = {}        # set name -> completion event (GPU) / work handle (CPU)
        self.time_waits = False   # bench.py: record an event pair around every wait on the compute stream
        self.wait_events = []
        self._rehearsal = None    # (workgroups, threads, passes) of the contention probe
        self._scratch = None
        self.reserved_cus = 0
        if self.enabled and self.on_gpu:
            env = os.environ.get("LG_RESERVED_CUS")
            self.reserve_cus(int(env) if env not in (None, "") else RESERVED_CUS_DP)

    def force_enable(self):
        """Run the exchange even at world size 1 (a one-GPU box: librccl loads, the communicator is built, the side-stream / event
        plumbing and the per-set waits are the real ones; an all-reduce over one rank leaves the gradients as they are)."""
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradSync.force_enable: no process group")
        self.enabled = True
        self.world_size = dist.get_world_size()
        if self.on_gpu and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.device)

    # ------------------------------------------------------------------ CU budget / rehearsal
    def reserve_cus(self, n: int):
        """Persistent kernels size their grids to (CUs - n) from now on (process-wide)."""
        from . import _lib
        _lib.check(_lib.load().lg_set_reserved_cus(int(n)), "lg_set_reserved_cus")
        self.reserved_cus = int(n)

    def rehearse(self, workgroups: int, threads: int = 512, passes: int = 2):
        """One-GPU rehearsal: every `launch` enqueues `workgroups` streaming read-add-write blocks over a scratch range of
        the all-reduce's size on a side stream (0 = off).  Gradients are untouched."""
        if not self.on_gpu:
            raise ValueError("rehearse: needs a GPU")
        if workgroups <= 0:
            self._rehearsal = None
            return
        self._rehearsal = (int(workgroups), int(threads), int(passes))
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.device)

    def _side_event(self, fn):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            fn()
            done = torch.cuda.Event()
            done.record(self.comm_stream)
        return done

    # ------------------------------------------------------------------ exchange
    def launch(self, name, store, start, end):
        """All-reduce(sum) store.grad[start:end]; non-blocking for the compute stream."""
        if name in self._pending:   # a second launch would drop the first one's event / work handle (on gloo: a handle nobody waits for)
            raise RuntimeError(f"GradSync.launch: the exchange of set {name!r} is still pending — wait({name!r}) first")
        if self._rehearsal is not None:
            from . import _lib
            n = (end - start) // 4 * 4
            if self._scratch is None or self._scratch.shape[1] < n:
                self._scratch = torch.zeros(2, max(n, store.grad.numel()), dtype=torch.float32, device=self.device)
            wg, th, ps = self._rehearsal
            lib = _lib.load()
            a, b = self._scratch[0], self._scratch[1]
            self._pending[name] = self._side_event(lambda: _lib.check(
                lib.lg_contention_probe(a.data_ptr(), b.data_ptr(), n, wg, th, ps, torch.cuda.current_stream(self.device).cuda_stream),
                "lg_contention_probe"))
            return
        if not self.enabled:
            return
        buf = store.grad[start:end]
        if self.on_gpu:
            self._pending[name] = self._side_event(lambda: dist.all_reduce(buf, op=dist.ReduceOp.SUM))
        else:
            self._pending[name] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)

    def _wait(self, names):
        todo = [(n, self._pending.pop(n)) for n in names if n in self._pending]
        timed = self.time_waits and self.on_gpu and todo
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        for _, p in todo:
            if self.on_gpu:
                torch.cuda.current_stream(self.device).wait_event(p)
            else:
                p.wait()
        if timed:   # e1 - e0 = how long the compute stream sat waiting for gradients still on the wire (0 = fully overlapped)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            self.wait_events.append((e0, e1))

    def wait(self, name):
        """Make the compute stream (or the host, on CPU) wait for the all-reduce of ONE gradient set."""
        self._wait([name])

    def wait_all(self):
        """... for every outstanding all-reduce."""
        self._wait(list(self._pending))
