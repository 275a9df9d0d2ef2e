# Round 5: conv_down3's item lists — a contiguous item range per XCD against the round-2 lists lb, lb + G, ... (LG_D3_OLD_LISTS=1)
mkdir -p gpurun_out/r5o
for b in 256 512; do for ol in 1 "" 1 ""; do
  echo "B=$b old_lists=${ol:-0}"; LG_B=$b LG_D3_OLD_LISTS=$ol timeout -k 10 120 python scripts/bench_layer.py "conv4 fwd" "convT1 dgrad" "conv2 fwd" "convT4 dgrad" 2>&1 | grep -v amdgpu.ids
done; done
