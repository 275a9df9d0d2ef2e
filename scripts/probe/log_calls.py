import sys, os
sys.path.insert(0, "/root/repo")
import torch, collections
import bench
from littlegan_amd import ops
from littlegan_amd.eager_trainer import EagerTrainer
args = bench.make_args("c3", "cuda:0")
from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator
decoder, encoder = Decoder(args), Encoder(args)
gen = Generator(args, decoder)
disc = Discriminator(args, encoder)
adj = Adjuster(args, disc, gen)
tr = EagerTrainer(args, gen, disc, adj, None)
inp = bench.synthetic_inputs(args, "cuda:0", 0)
log = collections.Counter()
for name in ("instnorm_apply", "instnorm_bwd", "conv2d_s2_fwd_stats", "convT_s2_fwd_stats", "conv2d_s2_wgrad", "convT_s2_wgrad", "convT_s1_tanh_fwd", "conv2d_s2_dgrad", "convT_s2_dgrad"):
    orig = getattr(ops, name)
    def mk(orig, name):
        def f(*a, **k):
            shp = None
            for t in list(a) + list(k.values()):
                if torch.is_tensor(t) and t.dim() == 4:
                    shp = tuple(t.shape); break
            extra = ""
            if name == "instnorm_apply":
                extra = f" skip={None if a[2] is None else a[2].dtype} want_f32={k.get('want_f32', True)} out16={k.get('out16') is not None}"
            log[(name, shp, extra)] += 1
            return orig(*a, **k)
        return f
    setattr(ops, name, mk(orig, name))
import littlegan_amd.model as M
tr.train_step_from_inputs(12, inp)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: (kv[0][0], -(kv[0][1][0]*kv[0][1][1]*kv[0][1][2]*kv[0][1][3] if kv[0][1] else 0))):
    print(v, k)
