# A/B on one box: conv_up3 fragment step pinned in three groups (0) vs interleaved by sched_group_barrier (1)
cd $GRAFT_REPO_ROOT
export LG_ALLOW_PROBE_BUILD=1   # _lib.load() refuses an ablation build otherwise
trap 'env -u LG_EXTRA_FLAGS python -m littlegan_amd.csrc.build > /dev/null 2>&1' EXIT   # leave the DEFAULT build in place
for sc in 0 1; do
  touch littlegan_amd/csrc/conv_up3.hip; LG_EXTRA_FLAGS="-DLG_U3_SCHED=$sc" python -m littlegan_amd.csrc.build > /dev/null 2>&1
  echo "SCHED=$sc"; timeout -k 10 100 python scripts/bench_layer.py "convT3 fwd" "convT4 fwd" "conv2 dgrad"; LG_FUSE=1 timeout -k 10 100 python scripts/bench_layer.py "conv2 dgrad"
done
