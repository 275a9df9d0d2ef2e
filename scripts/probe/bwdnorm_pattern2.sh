set -x
mkdir -p gpurun_out/r5b
for v in d3chk d3rfl d3wait; do
  chk=""; [ $v = d3chk ] && chk=1
  LG_CHK=$chk LG_LIB_VARIANT=$v timeout -k 10 120 python tests/diagnostics/bwdnorm_pattern.py > gpurun_out/r5b/pattern_$v.log 2>&1 || echo "FAILED $v"
done
LG_D3_STAGGER=0 LG_LIB_VARIANT=d3plain timeout -k 10 120 python tests/diagnostics/bwdnorm_pattern.py > gpurun_out/r5b/pattern_d3plain_nostagger.log 2>&1 || echo FAILED nostagger
grep -c "differing elements [1-9]" gpurun_out/r5b/*.log
