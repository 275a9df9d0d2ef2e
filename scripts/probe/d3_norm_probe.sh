# probe: conv_down3 with the InstanceNorm + LeakyReLU arithmetic added to its halo staging (LG_D3_DBG=16; results wrong, timing only)
set -e
cd ${GRAFT_REPO_ROOT:-.}
# Ablation variants are libraries of their OWN (liblittlegan_hip_<variant>.so, csrc/build_<variant>/): the product library is never
# rebuilt or replaced.  Build the variants in the dev container first (same command; the .so files travel with gpurun) — on the GPU
# box `variant` then finds them up to date and only runs.
variant() { [ -f "littlegan_amd/liblittlegan_hip_$1.so" ] || LG_EXTRA_FLAGS="$2" python -m littlegan_amd.csrc.build --variant "$1" > /dev/null 2>&1 || { echo "variant $1 failed to build"; exit 1; }; export LG_LIB_VARIANT="$1"; }
for d in 0 16; do
  variant "d3_dbg_$d" "-DLG_D3_DBG=$d"
  echo "DBG=$d"; timeout -k 10 100 python scripts/bench_layer.py "conv2 fwd" "conv3 fwd" "conv4 fwd"
done
