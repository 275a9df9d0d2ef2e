"""CLI mirror of /root/reference/main.py on the HIP hot path:  python main.py <mode> <exp_name> [-e env] [-g gpus] [--debug]
Modes kept: train, random-sample, condition-sample, evaluate-sample, evaluate, export-model.  `evaluate` runs the FID
arithmetic (main.py:82-104 -> evaluate.py calc) on SAVED Inception activations of the evaluate-sample images: the frozen
Inception graph the reference downloads cannot be obtained here.  `visual` (tensorboard) and `plot` (pydot) are UI tooling
outside the hot path."""
import os
import time

import numpy as np
import torch

from littlegan_amd.config import Arg

# configuration files are looked up next to this script (the reference reads them from the working directory);
# LITTLEGAN_CONFIG_DIR points somewhere else (tests)
args = Arg(config_dir=os.environ.get("LITTLEGAN_CONFIG_DIR", os.path.dirname(os.path.abspath(__file__))))

from littlegan_amd.dataset import CelebA
from littlegan_amd.eager_trainer import EagerTrainer
from littlegan_amd.model import Adjuster, Decoder, Discriminator, Encoder, Generator
from littlegan_amd.utils import save_image

if "LOCAL_RANK" in os.environ:  # one process per GPU (torchrun); the reference's -g only set CUDA_VISIBLE_DEVICES
    import torch.distributed as dist
    if os.environ.get("LITTLEGAN_DP_BACKEND", "nccl") == "gloo":  # rehearsal: ranks share cuda:0, gloo moves the tensors
        args.device = "cuda:0"
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        args.device = f"cuda:{os.environ['LOCAL_RANK']}"
        dist.init_process_group("nccl", device_id=torch.device(args.device))
print("Application Params: ", args)
print("Running Mode:", args.mode)
print(" - Initializing Networks...")
decoder = Decoder(args)
encoder = Encoder(args)
generator = Generator(args, decoder)
discriminator = Discriminator(args, encoder)
adjuster = Adjuster(args, discriminator, generator)
path = os.path

if args.mode == "train":
    data = CelebA(args)
    print("Using Attribute:", data.label)
    model = EagerTrainer(args, generator, discriminator, adjuster, data)
    model.train()
elif args.mode == "random-sample":
    args.reuse = True
    data = CelebA(args)
    model = EagerTrainer(args, generator, discriminator, adjuster, data)
    iterator = data.get_new_iterator()
    now_time = int(time.time())
    for b in range(args.random_sample_batch):
        image, cond = iterator.get_next()
        noise = torch.randn(cond.shape[0], args.noise_dim, device=cond.device)
        model.predict(noise, cond, image,
                      path.join(args.result_dir, "sample", "generator-%s-%d.jpg" % (now_time, b)),
                      path.join(args.result_dir, "sample", "discriminator-%s-%d.json" % (now_time, b)),
                      path.join(args.result_dir, "sample", "adjuster-%s-%d.jpg" % (now_time, b)))
        np.savez_compressed(path.join(args.result_dir, "sample", "input_data-%s-%d.npz" % (now_time, b)),
                            n=noise.cpu().numpy(), c=cond.cpu().numpy(), i=image.cpu().numpy())
elif args.mode == "evaluate-sample":
    args.reuse = True
    data = CelebA(args)
    model = EagerTrainer(args, generator, discriminator, adjuster, data)
    iterator = data.get_new_iterator()
    batches = int(np.ceil(args.evaluate_sample_size / args.batch_size))
    for b in range(min(batches, data.batches)):
        base_index = b * args.batch_size + 1
        image, cond = iterator.get_next()
        noise = torch.randn(cond.shape[0], args.noise_dim, device=cond.device)
        gen_image, save, adj_real, adj_fake = model.predict(noise, cond, image, None,
                                                            path.join(args.result_dir, "evaluate", "disc", str(b) + ".json"), None)
        for i in range(args.batch_size):
            save_image(gen_image[i], path.join(args.result_dir, "evaluate", "gen", str(base_index + i) + ".jpg"))
            if adj_real is not None and adj_fake is not None:
                save_image(adj_real[i], path.join(args.result_dir, "evaluate", "adj", "real_" + str(base_index + i) + ".jpg"))
                save_image(adj_fake[i], path.join(args.result_dir, "evaluate", "adj", "fake_" + str(base_index + i) + ".jpg"))
elif args.mode == "evaluate":
    # main.py:82-104: FID of the evaluate-sample images (gen, and adj when the Adjuster is trained) against the pre-calculated
    # statistics <test_data_dir>/<evaluate_pre_calculated>; one log line per call in evaluate/fid-{gen,adj}.log
    from littlegan_amd import fid
    for kind in ["gen"] + (["adj"] if args.train_adj else []):
        print("Running: \"evaluate calc %s\"" % kind)
        fid.calc(path.join(args.result_dir, "evaluate", kind), path.join(args.test_data_dir, args.evaluate_pre_calculated),
                 path.join(args.result_dir, "evaluate", "fid-%s.log" % kind))
elif args.mode == "condition-sample":
    args.reuse = True
    model = EagerTrainer(args, generator, discriminator, adjuster, None)
    cond = torch.tensor([[0., 0., 0., 0., 0., 1., 0.], [0., 0., 0., 0., 0., 1., 1.], [0., 0., 0., 0., 0., 0., 1.],
                         [1., 0., 0., 0., 0., 0., 1.], [1., 0., 0., 0., 1., 0., 1.], [1., 0., 1., 0., 1., 0., 1.],
                         [1., 1., 1., 0., 1., 0., 1.], [1., 1., 1., 1., 1., 0., 1.]], device=model.device)
    for i in range(1, 1 + args.condition_sample_batch):
        noise = torch.randn(1, args.noise_dim, device=model.device).repeat(8, 1)
        img = model.generator([noise, cond])
        save_image(img, path.join(args.result_dir, "sample", "condition-gen-%d.jpg" % i), (1, 8))
elif args.mode == "export-model":
    args.reuse = True
    args.restore = True
    model = EagerTrainer(args, generator, discriminator, adjuster, None)
    print("exported", model.export_model_checkpoint())
else:
    raise SystemExit(f"mode '{args.mode}' (tensorboard / pydot / FID tooling) is outside the hot path of this build")
