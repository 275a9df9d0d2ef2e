set -x
cd /root/repo && mkdir -p gpurun_out/r5a
for v in d3plain d3plainnp ""; do
  for bpc in 2 1; do
    LG_LIB_VARIANT=$v LG_D3_BLOCKS_PER_CU=$bpc timeout -k 10 120 python tests/diagnostics/bwdnorm_pattern.py > gpurun_out/r5a/pattern_${v:-product}_bpc$bpc.log 2>&1 || echo "FAILED $v $bpc"
  done
done
LG_LIB_VARIANT=d3plain LG_B=64 timeout -k 10 120 python tests/diagnostics/bwdnorm_pattern.py > gpurun_out/r5a/pattern_d3plain_B64.log 2>&1 || echo FAILED B64
