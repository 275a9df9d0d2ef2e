cd $GRAFT_REPO_ROOT
export LG_ALLOW_PROBE_BUILD=1   # _lib.load() refuses an ablation build otherwise
trap 'env -u LG_EXTRA_FLAGS python -m littlegan_amd.csrc.build > /dev/null 2>&1' EXIT   # leave the DEFAULT build in place
touch littlegan_amd/csrc/conv_down3.hip; LG_EXTRA_FLAGS="-DLG_D3_STAMPS" python -m littlegan_amd.csrc.build > /dev/null 2>&1
python scripts/probe/d3_stamps.py
