# A/B on one box: conv_down3 weight-ring depth / A-fragment prefetch depth (LG_D3_RING, LG_D3_ADEPTH) with the interleaved schedule
cd ${GRAFT_REPO_ROOT:-.}
# Ablation variants are libraries of their OWN (liblittlegan_hip_<variant>.so, csrc/build_<variant>/): the product library is never
# rebuilt or replaced.  Build the variants in the dev container first (same command; the .so files travel with gpurun) — on the GPU
# box `variant` then finds them up to date and only runs.
variant() { [ -f "littlegan_amd/liblittlegan_hip_$1.so" ] || LG_EXTRA_FLAGS="$2" python -m littlegan_amd.csrc.build --variant "$1" > /dev/null 2>&1 || { echo "variant $1 failed to build"; exit 1; }; export LG_LIB_VARIANT="$1"; }
for cfg in "10 2" "10 3" "5 3"; do
  set -- $cfg
  variant "d3_ring_$1_d3_adepth_$2" "-DLG_D3_RING=$1 -DLG_D3_ADEPTH=$2"
  echo "RING=$1 ADEPTH=$2"; timeout -k 10 100 python scripts/bench_layer.py "conv2 fwd" "conv3 fwd" "conv4 fwd" "convT4 dgrad" "convT3 dgrad" "convT2 dgrad"
done
