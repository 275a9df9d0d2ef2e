# residency / clock / phase census of conv_down3: see scripts/probe/d3_census.py
cd ${GRAFT_REPO_ROOT:-.}
variant() { [ -f "littlegan_amd/liblittlegan_hip_$1.so" ] || LG_EXTRA_FLAGS="$2" python -m littlegan_amd.csrc.build --variant "$1" > /dev/null 2>&1 || { echo "variant $1 failed to build"; exit 1; }; export LG_LIB_VARIANT="$1"; }
variant d3stamps "-DLG_D3_STAMPS"
for l in conv2 conv3; do
  echo "=== $l, two blocks per CU, all stamps"; timeout -k 10 200 python scripts/probe/d3_census.py $l || exit 1
  echo "=== $l, two blocks per CU, first / last stamp only"; LG_D3_STAMPS_LITE=1 timeout -k 10 200 python scripts/probe/d3_census.py $l || exit 1
  echo "=== $l, ONE block per CU (a lone wave per SIMD), all stamps"; LG_D3_BLOCKS_PER_CU=1 timeout -k 10 200 python scripts/probe/d3_census.py $l || exit 1
done
