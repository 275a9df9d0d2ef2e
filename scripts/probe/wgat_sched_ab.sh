# A/B on one box: wgrad_at k step in pinned groups (0) vs interleaved by sched_group_barrier (1)
cd $GRAFT_REPO_ROOT
export LG_ALLOW_PROBE_BUILD=1   # _lib.load() refuses an ablation build otherwise
trap 'env -u LG_EXTRA_FLAGS python -m littlegan_amd.csrc.build > /dev/null 2>&1' EXIT   # leave the DEFAULT build in place
for sc in 0 1; do
  touch littlegan_amd/csrc/wgrad_at.hip; LG_EXTRA_FLAGS="-DLG_WGAT_SCHED=$sc" python -m littlegan_amd.csrc.build > /dev/null 2>&1
  echo "SCHED=$sc"; timeout -k 10 100 python scripts/bench_layer.py wgrad
done
