# Round 5: convT4 forward (conv_up3<64,32>, one tile per step) — two (class, pixel half) units per wave (13 / 13 / 12 / 12 tap units) against one rotating
# class per wave (9 / 6 / 6 / 4): parity first, then the stack bench, A/B by LG_U3_NO_HALVES
mkdir -p gpurun_out/r5v
timeout -k 10 600 python -m pytest tests/test_launch_shapes_gpu.py tests/test_ops_gpu.py tests/test_step_replay_gpu.py -m gpu -x -q -k "dec.conv4 or convT or up3 or persistent or forward or whole or replay" > gpurun_out/r5v/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r5v/tests.log
for nh in 1 "" 1 ""; do
  LG_U3_NO_HALVES=$nh timeout -k 10 200 python scripts/bench_gstack.py gpurun_out/r5v/gstack_nohalves${nh:-0}_$RANDOM.json > /dev/null 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5v/gstack_*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], ' | '.join(f"{l['layer'].split()[0]} {l['us_median']:.1f} ({l['kernel']})" if 'T4' in l['layer'] else f"{l['layer'].split()[0]} {l['us_median']:.1f}" for l in d['layers']), '| total', d.get('total_us_median'))
PY
