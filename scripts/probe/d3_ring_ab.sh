# A/B on one box: conv_down3 weight-ring depth / A-fragment prefetch depth (LG_D3_RING, LG_D3_ADEPTH) with the interleaved schedule
cd $GRAFT_REPO_ROOT
export LG_ALLOW_PROBE_BUILD=1   # _lib.load() refuses an ablation build otherwise
trap 'env -u LG_EXTRA_FLAGS python -m littlegan_amd.csrc.build > /dev/null 2>&1' EXIT   # leave the DEFAULT build in place
for cfg in "10 2" "10 3" "5 3"; do
  set -- $cfg
  touch littlegan_amd/csrc/conv_down3.hip; LG_EXTRA_FLAGS="-DLG_D3_RING=$1 -DLG_D3_ADEPTH=$2" python -m littlegan_amd.csrc.build > /dev/null 2>&1
  echo "RING=$1 ADEPTH=$2"; timeout -k 10 100 python scripts/bench_layer.py "conv2 fwd" "conv3 fwd" "conv4 fwd" "convT4 dgrad" "convT3 dgrad" "convT2 dgrad"
done
