# A/B on one box: conv_down3 tap body pinned in three groups (0) vs interleaved by sched_group_barrier (1)
cd ${GRAFT_REPO_ROOT:-.}
# Ablation variants are libraries of their OWN (liblittlegan_hip_<variant>.so, csrc/build_<variant>/): the product library is never
# rebuilt or replaced.  Build the variants in the dev container first (same command; the .so files travel with gpurun) — on the GPU
# box `variant` then finds them up to date and only runs.
variant() { [ -f "littlegan_amd/liblittlegan_hip_$1.so" ] || LG_EXTRA_FLAGS="$2" python -m littlegan_amd.csrc.build --variant "$1" > /dev/null 2>&1 || { echo "variant $1 failed to build"; exit 1; }; export LG_LIB_VARIANT="$1"; }
for sc in 0 1; do
  variant "d3_sched_$sc" "-DLG_D3_SCHED=$sc"
  echo "SCHED=$sc"; timeout -k 10 100 python scripts/bench_layer.py "conv2 fwd" "conv3 fwd" "conv4 fwd" "convT4 dgrad" "convT3 dgrad" "convT2 dgrad"
done
