# Round 5: the adopted exact-f32 tiling rules (split DOWN contraction on small 8x8 grids, 128-column UP tiles on full ones, class order)
# against each switched off, per layer at B = 64 / 128 and in the C2 step.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
L=gpurun_out/r5x/f32_final_ab.log
: > $L
for B in 64 128; do for v in "LG_X=0" "LG_NO_F32_KSPLIT=1 LG_NO_F32_UPWIDE=1 LG_NO_UP_CLSORDER=1"; do
  echo "== B=$B $v" >> $L
  env $v LG_DT=f32 LG_B=$B timeout -k 10 120 python scripts/bench_layer.py fwd 2>&1 | grep -v amdgpu.ids >> $L || exit 1
done; done
for v in "LG_X=0" "LG_NO_F32_KSPLIT=1" "LG_NO_F32_UPWIDE=1" "LG_NO_UP_CLSORDER=1" "LG_NO_F32_KSPLIT=1 LG_NO_F32_UPWIDE=1 LG_NO_UP_CLSORDER=1" "LG_X=0"; do
  echo "== C2 step $v" >> $L
  env $v timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $L || exit 1
done
cat $L
