# Round 5: block -> tile order of conv_halo on the exact-f32 path: column tile fastest (LG_HALO_MFAST unset) against row tile fastest
# where the weights exceed 2 MB (=1) or always (=2); with and without the split contraction's 64-column form.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
L=gpurun_out/r5x/f32_mfast_ab.log
: > $L
for B in 64 128; do for v in "LG_HALO_MFAST=0" "LG_HALO_MFAST=1" "LG_HALO_MFAST=2" "LG_HALO_MFAST=1 LG_F32_KSPLIT_BN64=1"; do
  echo "== B=$B $v" >> $L
  env $v LG_DT=f32 LG_B=$B timeout -k 10 120 python scripts/bench_layer.py fwd 2>&1 | grep -v amdgpu.ids >> $L || exit 1
done; done
for v in "LG_HALO_MFAST=0" "LG_HALO_MFAST=1" "LG_HALO_MFAST=2" "LG_HALO_MFAST=1 LG_F32_KSPLIT_BN64=1" "LG_HALO_MFAST=0"; do
  echo "== C2 step $v" >> $L
  env $v timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $L || exit 1
done
cat $L
