"""Diagnostic: where does the HOST time of an eager step go (cProfile over 20 steps of the C3 configuration)?"""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from littlegan_amd.eager_trainer import EagerTrainer
from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator
args = bench.make_args("c3", "cuda:0")
dec, enc = Decoder(args), Encoder(args)
g = Generator(args, dec); d = Discriminator(args, enc); a = Adjuster(args, d, g)
tr = EagerTrainer(args, g, d, a, None)
inp = bench.synthetic_inputs(args, torch.device("cuda:0"), 0)
for i in range(15):
    tr.train_step_from_inputs(11 + i, inp)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.train_step_from_inputs(26 + i, inp)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
