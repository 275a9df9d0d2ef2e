set -e
cd $GRAFT_REPO_ROOT
export LG_ALLOW_PROBE_BUILD=1   # _lib.load() refuses an ablation build otherwise
trap 'env -u LG_EXTRA_FLAGS python -m littlegan_amd.csrc.build > /dev/null 2>&1' EXIT   # leave the DEFAULT build in place
for d in 0 1 2 4 3 6; do
  touch littlegan_amd/csrc/wgrad_at.hip; LG_EXTRA_FLAGS="-DLG_WGAT_DBG=$d" python -m littlegan_amd.csrc.build > /dev/null 2>&1
  echo "DBG=$d"; timeout -k 10 100 python scripts/bench_layer.py "wgrad 32/64" "wgrad 64/128"
done
