cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== no pair (conv_halo) rep $rep"
  LG_NO_UP4_PAIR=1 python scripts/bench_layer.py "convT1 fwd" "conv4 dgrad" 2>&1 | grep -v amdgpu.ids
  LG_NO_UP4_PAIR=1 LG_FUSE=1 python scripts/bench_layer.py "conv4 dgrad" 2>&1 | grep -v amdgpu.ids
  echo "== pair tiles (conv_up4) rep $rep"
  python scripts/bench_layer.py "convT1 fwd" "conv4 dgrad" 2>&1 | grep -v amdgpu.ids
  LG_FUSE=1 python scripts/bench_layer.py "conv4 dgrad" 2>&1 | grep -v amdgpu.ids
done
LG_B=512 LG_NO_UP4_PAIR=1 python scripts/bench_layer.py "convT1 fwd" "conv4 dgrad" 2>&1 | grep -v amdgpu.ids
LG_B=512 python scripts/bench_layer.py "convT1 fwd" "conv4 dgrad" 2>&1 | grep -v amdgpu.ids
echo "== bench no pair"; LG_NO_UP4_PAIR=1 python bench.py --no-cpu-baseline --no-graph-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
echo "== bench pair"; python bench.py --no-cpu-baseline --no-graph-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
echo "== bench no pair"; LG_NO_UP4_PAIR=1 python bench.py --no-cpu-baseline --no-graph-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
echo "== bench pair"; python bench.py --no-cpu-baseline --no-graph-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
