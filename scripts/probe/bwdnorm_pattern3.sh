set -x
mkdir -p gpurun_out/r5c
for v in d3delay d3vmov; do
  LG_LIB_VARIANT=$v timeout -k 10 120 python tests/diagnostics/bwdnorm_pattern.py > gpurun_out/r5c/pattern_$v.log 2>&1 || echo "FAILED $v"
done
grep -c "differing elements [1-9]" gpurun_out/r5c/*.log
