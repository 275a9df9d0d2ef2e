"""Diagnostic: what do the per-launch HIP events of bench.py's roofline leg (ops.Profile) cost the eager step?  Interleaved rounds in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from littlegan_amd import ops
from littlegan_amd.eager_trainer import EagerTrainer
from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator
args = bench.make_args("c3", "cuda:0")
dec, enc = Decoder(args), Encoder(args)
g = Generator(args, dec); d = Discriminator(args, enc); a = Adjuster(args, d, g)
tr = EagerTrainer(args, g, d, a, None)
inp = bench.synthetic_inputs(args, torch.device("cuda:0"), 0)
b = 11
for i in range(15):
    tr.train_step_from_inputs(b, inp); b += 1
torch.cuda.synchronize()
def run(n, events):
    global b
    if events: ops.Profile.start()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        tr.train_step_from_inputs(b, inp); b += 1
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    if events: ops.Profile.stop()
    return dt
for r in range(4):
    print(f"round {r}: with events {run(30, True):.3f} ms/step   without {run(30, False):.3f} ms/step")
