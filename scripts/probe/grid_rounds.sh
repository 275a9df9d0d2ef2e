# Round 5: wave quantisation of the C3 and C2 steps' launches (scripts/grid_rounds.py over one --pmc pass each)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
cd /tmp && export TMPDIR=/tmp
for w in c3 c2; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5q/$w -o g -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 1 --warmup 11 --no-cpu-baseline --no-graph-leg > $GRAFT_REPO_ROOT/gpurun_out/r5q/$w.log 2>&1 || exit 1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r5q/$w -name "*counter_collection.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/grid_rounds.py $f > $GRAFT_REPO_ROOT/gpurun_out/r5q/${w}_grid_rounds.md
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r5q/$w
  echo "== $w"; cat $GRAFT_REPO_ROOT/gpurun_out/r5q/${w}_grid_rounds.md
done
