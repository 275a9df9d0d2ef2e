"""The CPU restatement (oracle/torch_oracle.py, torch-CPU fp32 — NOT TensorFlow, which cannot run in this pipeline) timed ONCE at the
CONFIGURED batches on the GPU box's host cores: C3 (128x128, B = 256, G + D + Adjuster) and C2 (128x128, B = 64, G + D only).
bench.py's `cpu_baseline` field times a bounded B = 8 sample of the same step; this is the unbounded companion (minutes).

usage: python scripts/cpu_baseline_full.py out.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import _cpu_time_step  # noqa: E402

out = {"what": "torch-CPU fp32 restatement of the training step at the configured batches (test infrastructure timed as a baseline)",
       "threads": torch.get_num_threads(), "nproc": os.cpu_count()}
for name, kw, B in (("C2 (128x128, B=64, G+D)", dict(train_adj=False, init_dim=8), 64), ("C3 (128x128, B=256, G+D+Adj)", dict(train_adj=True, init_dim=8), 256)):
    t0 = time.time()
    v, n, dt = _cpu_time_step(kw, B, 2, 0.0, max_steps=2)
    out[name] = {"images_per_sec": round(v, 3), "timed_steps": n, "s_per_step": round(dt, 2), "warmup_steps": 3, "wall_s": round(time.time() - t0, 1)}
    print(name, out[name], flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
