# Round 5: conv_up4's item order, row tile fastest + XCD-major ranks (LG_U4_MFAST=1: an XCD's blocks share ONE column tile's weights, 2.4 MB of the
# 4.9 MB of convT1's weights per L2) against column tile fastest (the default: every XCD streams all of them)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5u
for b in 256 512; do for x in "" 1 "" 1; do echo "B=$b mfast=${x:-0}"; LG_B=$b LG_U4_MFAST=$x timeout -k 10 120 python scripts/bench_layer.py "convT1 fwd" "conv4 dgrad" 2>&1 | grep -v amdgpu.ids; done; done
for x in "" 1 "" 1; do
  LG_U4_MFAST=$x timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r5u/bench_m${x:-0}.json 2>/dev/null || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/r5u/bench_m${x:-0}.json').read().strip().splitlines()[-1]); print('mfast=${x:-0}', d['ms_per_step'], d['value'], [ (k, v['ms_per_step']) for k, v in d['roofline']['all_kernels'].items() if 'conv_up4' in k])"
done
LG_U4_MFAST=1 timeout -k 10 400 python -m pytest tests/test_launch_shapes_gpu.py -m gpu -x -q -k "dec.conv1 or enc.conv4" 2>&1 | tail -2
