# Round 5: norm-pass grids cut to whole rounds of their kernel's residency (default) against the caps alone (LG_NO_NORM_ROUNDS=1): passes in isolation, C3 and C2 steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
L=gpurun_out/r5n/norm_rounds_ab.log
: > $L
for v in "LG_X=0" "LG_NO_NORM_ROUNDS=1"; do
  echo "== $v" >> $L
  env $v timeout -k 10 200 python scripts/bench_norm.py 2>&1 | grep -v amdgpu.ids | grep "bwd+db\|bwd  " >> $L || exit 1
done
for v in "LG_X=0" "LG_NO_NORM_ROUNDS=1" "LG_X=0" "LG_NO_NORM_ROUNDS=1" "LG_X=0" "LG_NO_NORM_ROUNDS=1"; do
  echo -n "C3 step $v " >> $L
  env $v timeout -k 10 300 python bench.py --no-cpu-baseline --no-graph-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $L || exit 1
done
for v in "LG_X=0" "LG_NO_NORM_ROUNDS=1" "LG_X=0" "LG_NO_NORM_ROUNDS=1"; do
  echo -n "C2 step $v " >> $L
  env $v timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $L || exit 1
done
cat $L
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "norm or step or whole" > gpurun_out/r5n/tests.log 2>&1; tail -3 gpurun_out/r5n/tests.log
