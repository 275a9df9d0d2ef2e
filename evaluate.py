"""CLI mirror of /root/reference/evaluate.py:  python evaluate.py {pre-calculate|calc} image_path stats_path model_path output_file
The arithmetic is the reference's (fid.py:112-163,185-188) with the activation mean / covariance on the MI355X (lg_fid_stats).
`image_path` names the saved Inception pool_3 activations of the images (an .npy / .npz file, or a directory holding
activations.npy) instead of the JPEGs themselves: the frozen Inception graph the reference downloads (fid.py:276) cannot be
obtained here; `model_path` (where the reference keeps that graph) is accepted and unused."""
import argparse

from littlegan_amd import fid

parser = argparse.ArgumentParser()
parser.add_argument("mode", choices=["pre-calculate", "calc"])
parser.add_argument("image_path")
parser.add_argument("stats_path")
parser.add_argument("model_path")
parser.add_argument("output_file", nargs="?", default=None)
parser.add_argument("--gpu", default="")
args = parser.parse_args()
if args.mode == "pre-calculate":
    fid.pre_calculate(args.image_path, args.stats_path)
else:
    if args.output_file is None:
        parser.error("calc needs an output log file")
    fid.calc(args.image_path, args.stats_path, args.output_file)
