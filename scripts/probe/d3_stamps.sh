cd $GRAFT_REPO_ROOT
touch littlegan_amd/csrc/conv_down3.hip; LG_EXTRA_FLAGS="-DLG_D3_STAMPS" python -m littlegan_amd.csrc.build > /dev/null 2>&1
python scripts/probe/d3_stamps.py
