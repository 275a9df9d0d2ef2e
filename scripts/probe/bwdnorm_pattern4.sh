set -x
mkdir -p gpurun_out/r5d
for v in d3pin0 d3pin1 d3pin4 d3pin16 d3plain; do
  LG_LIB_VARIANT=$v timeout -k 10 120 python tests/diagnostics/bwdnorm_pattern.py > gpurun_out/r5d/pattern_$v.log 2>&1 || echo "FAILED $v"
done
grep -c "differing elements [1-9]" gpurun_out/r5d/*.log
