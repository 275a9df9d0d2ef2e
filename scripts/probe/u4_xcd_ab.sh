# Round 5: conv_up4's block ranks inside a class-pair type, XCD-major (LG_U4_XCD=1) against plain blockIdx order
mkdir -p gpurun_out/r5u
for b in 256 512; do for x in "" 1; do echo "B=$b xcd=${x:-0}"; LG_B=$b LG_U4_XCD=$x timeout -k 10 120 python scripts/bench_layer.py "convT1 fwd" "convT2 fwd" "conv3 dgrad" "conv4 dgrad" 2>&1 | grep -v amdgpu.ids; done; done
for x in "" 1 "" 1; do
  LG_U4_XCD=$x timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r5u/bench_x${x:-0}.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r5u/bench_x${x:-0}.json').read().strip().splitlines()[-1]); print('xcd=${x:-0}', d['ms_per_step'], d['value'], [ (k, v['ms_per_step']) for k, v in d['roofline']['all_kernels'].items() if 'conv_up4' in k])"
done
