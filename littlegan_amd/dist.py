"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI; "gloo" on CPU for tests).  The reference has no distributed code at all; the step is
pure data parallel because InstanceNormalization has no batch statistics, so the ONLY exchange is the
mean of the three gradient sets.  Each set is one contiguous range of the flat gradient buffer
(model.ParamStore) => one large all-reduce per set (G 27 MB, D 18 MB, A 4 MB at 128^2), launched on a side
stream as soon as its tape's backward has been enqueued, overlapping the next tape's kernels.
The 1/world_size scale and the D-clip are applied afterwards inside the Adam kernel."""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, device):
        self.device = torch.device(device)
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world_size = dist.get_world_size() if self.enabled else 1
        self.on_gpu = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.enabled and self.on_gpu) else None
        self._pending = []
        self.time_waits = False   # bench.py: record an event pair around every wait_all() on the compute stream
        self.wait_events = []

    def launch(self, name, store, start, end):
        """All-reduce(sum) store.grad[start:end]; non-blocking for the compute stream."""
        if not self.enabled:
            return
        buf = store.grad[start:end]
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
            self._pending.append(done)
        else:
            self._pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))

    def wait_all(self):
        """Make the compute stream (or the host, on CPU) wait for every outstanding all-reduce."""
        timed = self.time_waits and self.on_gpu and self.enabled and self._pending
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        for p in self._pending:
            if self.on_gpu:
                torch.cuda.current_stream(self.device).wait_event(p)
            else:
                p.wait()
        if timed:   # e1 - e0 = how long the compute stream sat waiting for gradients still on the wire (0 = fully overlapped)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            self.wait_events.append((e0, e1))
        self._pending.clear()
