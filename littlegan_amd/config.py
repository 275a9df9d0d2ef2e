"""Mirror of /root/reference/config.py:6-42: sample.config.json < <env>.config.json < CLI, every key an
attribute, derived cond_dim / result_dir / gpu / prefetch.  Extra keys of this build: mfma_dtype ("f32" |
"bf16"), synthetic (bool: use the synthetic CelebA-shaped dataset), seed."""
import json
import os
from argparse import ArgumentParser

MODES = ["train", "plot", "visual", "random-sample", "evaluate", "condition-sample", "evaluate-sample", "export-model"]


class Arg:
    def __init__(self, argv=None, config_dir="."):
        print(" - Initializing Application...")
        parser = ArgumentParser(prog="LittleGAN", description="The code for paper: LittleGAN")
        parser.add_argument("mode", type=str, help="run mode", default="train", choices=MODES)
        parser.add_argument("exp_name", type=str, help="experience name")
        parser.add_argument("-e", "--env", type=str, help="config environment", default="sample")
        parser.add_argument("-g", "--gpu", type=str, required=False, help="gpu ids, eg: 0,1,2,3", default="-1")
        parser.add_argument("--debug", help="use debug mode, ignore git repo is dirty", action="store_true")
        args = parser.parse_args(argv)
        with open(os.path.join(config_dir, "sample.config.json")) as f:
            for k, v in json.load(f).items():
                setattr(self, k, v)
        self.env_file = args.env + ".config.json"
        with open(os.path.join(config_dir, self.env_file)) as f:
            for k, v in json.load(f).items():
                setattr(self, k, v)
        for k, v in vars(args).items():
            setattr(self, k, v)
        self.cond_dim = len(self.attr)
        self.result_dir = os.path.join(self.all_result_dir, args.exp_name)
        # the reference sets CUDA_VISIBLE_DEVICES (config.py:35); with one process per GPU the launcher
        # (torchrun) owns device selection, so only the parsed list is kept.
        self.gpu = [int(item) for item in self.gpu.split(",") if item.isnumeric() and int(item) >= 0]
        self.prefetch = self.prefetch_batch * self.batch_size

    def __str__(self):
        return self.__dict__.__str__()
