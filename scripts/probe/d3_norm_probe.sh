# probe: conv_down3 with the InstanceNorm + LeakyReLU arithmetic added to its halo staging (LG_D3_DBG=16; results wrong, timing only)
set -e
cd $GRAFT_REPO_ROOT
export LG_ALLOW_PROBE_BUILD=1   # _lib.load() refuses an ablation build otherwise
trap 'env -u LG_EXTRA_FLAGS python -m littlegan_amd.csrc.build > /dev/null 2>&1' EXIT   # leave the DEFAULT build in place
for d in 0 16; do
  touch littlegan_amd/csrc/conv_down3.hip; LG_EXTRA_FLAGS="-DLG_D3_DBG=$d" python -m littlegan_amd.csrc.build > /dev/null 2>&1
  echo "DBG=$d"; timeout -k 10 100 python scripts/bench_layer.py "conv2 fwd" "conv3 fwd" "conv4 fwd"
done
