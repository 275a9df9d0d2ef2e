# Round 5: convT4 forward (conv_up3<64,32>) — workgroups per CU of the form whose staging area aliases the halo (34 KB of LDS, 168 VGPRs, 4-deep
# weight ring, four barriers per step) against the round-4 form (61 KB, 210 VGPRs, 8-deep ring, rows leaving inside the next class loop)
mkdir -p gpurun_out/r5n
for cfg in "NO_ALIAS=1" "ALIAS_WGS=3" "ALIAS_WGS=2" "ALIAS_WGS=1" "NO_ALIAS=1" "ALIAS_WGS=3"; do
  env LG_U3_$cfg timeout -k 10 200 python scripts/bench_gstack.py gpurun_out/r5n/gstack_${cfg}_$RANDOM.json > /dev/null 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5n/gstack_*=*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], ' | '.join(f"{l['layer'].split()[0]} {l['us_median']:.1f} ({l['kernel']})" if 'T4' in l['layer'] else f"{l['layer'].split()[0]} {l['us_median']:.1f}" for l in d['layers']), '| total', d.get('total_us_median'))
PY
