"""Mirror of /root/reference/config.py:6-42: sample.config.json < <env>.config.json < CLI, every key an
attribute, derived cond_dim / result_dir / gpu / prefetch.  Extra keys of this build: mfma_dtype ("f32" |
"bf16"), synthetic (bool: use the synthetic CelebA-shaped dataset), seed."""
import json
import os
from argparse import ArgumentParser

# hyper-parameter defaults of the reference (sample.config.json:1-54), kept in code so that the JSON files of this
# build only carry overrides
DEFAULTS = {
    'batch_size': 32,
    'image_channel': 3,
    'image_path': '/path/to/image',
    'attr_path': '/path/to/attr/list.txt',
    'image_ext': 'jpg',
    'image_dim': 128,
    'attr': [8, 15, 20, 22, 26, 36, 39],
    'noise_dim': 93,
    'init_dim': 8,
    'norm': 'instance',
    'conv_filter': [384, 256, 128, 64, 32],
    'kernel_size': 5,
    'leaky_alpha': 0.3,
    'dropout_rate': 0.5,
    'l1_lambda': 0.02,
    'lr': 5e-05,
    'beta_1': 0.5,
    'beta_2': 0.9,
    'epoch': 100,
    'use_gp': False,
    'gp_weight': 5.0,
    'use_clip': True,
    'clip_range': 0.5,
    'use_partition': True,
    'partition_interval': 4,
    'freq_gen': 100,
    'freq_test': 2000,
    'all_result_dir': '/path/to/LittleGAN-result',
    'test_data_dir': '/path/to/LittleGAN-test',
    'evaluate_pre_calculated': 'fid_stats_celeba_128_all.npz',
    'random_sample_batch': 4,
    'condition_sample_batch': 100,
    'evaluate_sample_size': 30000,
    'restore': True,
    'reuse': False,
    'train_adj': True,
    'prefetch_batch': 3,
    'threads': 8,
    # keys added by this build
    'mfma_dtype': 'f32', 'synthetic': False, 'seed': 0,
}

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
        for k, v in DEFAULTS.items():
            setattr(self, k, v)
        self.env_file = args.env + ".config.json"
        for name in dict.fromkeys(["sample.config.json", self.env_file]):  # defaults < sample < env (config.py:19-29)
            fp = os.path.join(config_dir, name)
            if os.path.exists(fp):
                with open(fp) as f:
                    for k, v in json.load(f).items():
                        setattr(self, k, v)
            elif name != "sample.config.json":
                raise FileNotFoundError(fp)
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
