cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
L=gpurun_out/r5x/f32_ablate2.log
: > $L
for v in "" "LG_F32_KSPLIT_BN64=1" "LG_NO_F32_KSPLIT=1"; do for d in 0 1 2 4 6 7; do
  echo "== B=64 $v LG_DBG=$d" >> $L
  env $v LG_DT=f32 LG_B=64 LG_DBG=$d timeout -k 10 120 python scripts/bench_layer.py "conv4 fwd" 2>&1 | grep -v amdgpu.ids >> $L || exit 1
done; done
cat $L
