set -x
mkdir -p gpurun_out/r5f
for v in d3fz d3pad; do
  LG_LIB_VARIANT=$v timeout -k 10 200 python tests/diagnostics/bwdnorm_pattern.py > gpurun_out/r5f/pattern_$v.log 2>&1 || echo "FAILED $v"
done
grep -c "differing elements [1-9]" gpurun_out/r5f/*.log
