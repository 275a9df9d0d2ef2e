# Round 5: exact-f32 convT4 forward (N = 32) on the resident-halo form against the K-sliced one (LG_NO_F32_RES=1), per layer and in the C2 step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
L=gpurun_out/r5x/f32_res_ab.log
: > $L
for B in 64 128 256; do for v in "LG_X=0" "LG_NO_F32_RES=1"; do
  echo "== B=$B $v" >> $L
  env $v LG_DT=f32 LG_B=$B timeout -k 10 120 python scripts/bench_layer.py "convT4 fwd" 2>&1 | grep -v amdgpu.ids >> $L || exit 1
done; done
for v in "LG_X=0" "LG_NO_F32_RES=1" "LG_X=0" "LG_NO_F32_RES=1"; do
  echo "== C2 step $v" >> $L
  env $v timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $L || exit 1
done
cat $L
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "f32 or C2 or c2 or c1" > gpurun_out/r5x/f32_tests2.log 2>&1; tail -3 gpurun_out/r5x/f32_tests2.log
