# Round 5: wgrad_at's splits walking interleaved items (LG_WGAT_INTERLEAVE=1: the chip sweeps the operand maps front to back together) against contiguous item ranges per split
mkdir -p gpurun_out/r5t
for il in "" 1; do echo "interleave=${il:-0}"; LG_WGAT_INTERLEAVE=$il timeout -k 10 120 python scripts/bench_layer.py "wgrad" 2>&1 | grep -v amdgpu.ids; done
for il in "" 1 "" 1; do
  LG_WGAT_INTERLEAVE=$il timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r5t/bench_il${il:-0}.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r5t/bench_il${il:-0}.json').read().strip().splitlines()[-1]); print('interleave=${il:-0}', d['ms_per_step'], d['value'], [ (k, v['ms_per_step']) for k, v in d['roofline']['all_kernels'].items() if 'wgrad_at' in k])"
done
