cd ${GRAFT_REPO_ROOT:-.}
# Ablation variants are libraries of their OWN (liblittlegan_hip_<variant>.so, csrc/build_<variant>/): the product library is never
# rebuilt or replaced.  Build the variants in the dev container first (same command; the .so files travel with gpurun) — on the GPU
# box `variant` then finds them up to date and only runs.
variant() { [ -f "littlegan_amd/liblittlegan_hip_$1.so" ] || LG_EXTRA_FLAGS="$2" python -m littlegan_amd.csrc.build --variant "$1" > /dev/null 2>&1 || { echo "variant $1 failed to build"; exit 1; }; export LG_LIB_VARIANT="$1"; }
variant "d3_stamps" "-DLG_D3_STAMPS"
python scripts/probe/d3_stamps.py
