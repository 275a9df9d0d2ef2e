# Round 5: decoder level 4's weight gradient beside its data gradient on a second stream (LG_WG_SIDE=1, default) against one stream (=0), C3 step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
for v in 1 0 1 0 1 0; do
  echo -n "LG_WG_SIDE=$v "
  LG_WG_SIDE=$v timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['graph_replay']['ms_per_step'])" || exit 1
done | tee gpurun_out/r5o/wg_side_ab.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5o/gpu_suite.log 2>&1; tail -3 gpurun_out/r5o/gpu_suite.log
