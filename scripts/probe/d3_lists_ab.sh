# Round 5: conv_down3's item lists (LG_D3_LISTS): 0 = rounds of G items with the partial round dealt evenly over the XCDs (default),
# 1 = the round-2 lists lb, lb + G, ... (the remainder on the first XCDs), 2 = one contiguous item range per XCD
mkdir -p gpurun_out/r5o
for b in 512; do for ol in 1 0 2; do
  echo "B=$b lists=$ol"; LG_B=$b LG_D3_LISTS=$ol timeout -k 10 120 python scripts/bench_layer.py "conv4 fwd" "convT1 dgrad" "conv2 fwd" "convT4 dgrad" 2>&1 | grep -v amdgpu.ids
done; done
for ol in 1 0 2 1 0 2; do
  LG_D3_LISTS=$ol timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r5o/bench_lists$ol.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r5o/bench_lists$ol.json').read().strip().splitlines()[-1]); print('lists=$ol', d['ms_per_step'], d['value'], d['clock']['in_kernel_mhz'])"
done
