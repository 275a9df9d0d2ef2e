# Round 5: how much of conv_up3's time is the weight-fragment stream (a wave re-reads its class's 16 .. 36 KB of weights from L2 for EVERY tile)?
# Variant library u3nowl = conv_up3.hip built with -DLG_U3_NO_WLOAD (ring primed once, never refilled: results wrong, timing only) against the product.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5v
for v in "" u3nowl "" u3nowl; do
  LG_LIB_VARIANT=$v timeout -k 10 200 python scripts/bench_gstack.py gpurun_out/r5v/gstack_wl_${v:-product}_$RANDOM.json > /dev/null 2>&1 || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5v/gstack_wl_*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], ' | '.join(f"{l['layer'].split()[0]} {l['us_median']:.1f}" for l in d['layers']), '| total', d.get('total_us_median'))
PY
