# Round 5: where the exact-f32 8x8-level convs (conv4 forward 0.40, convT1 forward 0.56 of the fp32 matrix peak at B = 64) spend their time:
# conv_halo.hip's LG_DBG ablation bits (1 no MFMAs, 2 no weight-fragment loads, 4 no halo restage, 16 no stores, 32 no moments) — timing only.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
for B in 64 128; do for d in 0 1 2 4 16 6 7; do
  echo "== B=$B LG_DBG=$d"
  LG_DT=f32 LG_B=$B LG_DBG=$d timeout -k 10 120 python scripts/bench_layer.py "conv4 fwd" "convT1 fwd" "conv3 fwd" "convT2 fwd" 2>&1 | grep -v amdgpu.ids || exit 1
done; done > gpurun_out/r5x/f32_ablate.log 2>&1
cat gpurun_out/r5x/f32_ablate.log
