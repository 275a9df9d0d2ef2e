# Round 5: the non-deterministic BWDNORM build reproduced and read back.  Builds the reproducer variant of conv_down3.hip
# (per-item record by vector load, -DLG_D3_COEF_PLAIN) beside the product library and runs, on a GPU box: the launch-to-launch pattern
# (two and one block per CU), the one-hot-weight probe that reads the staged operand back, and the same two on the product library.
set -x
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/bwdnorm
# (the product library is built WITHOUT packed fp32 instructions since round 5: the reproducer switches them back on)
[ -f littlegan_amd/liblittlegan_hip_d3plain.so ] || LG_EXTRA_FLAGS="-DLG_D3_COEF_PLAIN -Xclang -target-feature -Xclang +packed-fp32-ops" LG_VARIANT_SOURCES=conv_down3.hip python -m littlegan_amd.csrc.build --variant d3plain
for v in d3plain ""; do
  for bpc in 2 1; do
    LG_LIB_VARIANT=$v LG_D3_BLOCKS_PER_CU=$bpc timeout -k 10 200 python tests/diagnostics/bwdnorm_pattern.py > gpurun_out/bwdnorm/pattern_${v:-product}_bpc$bpc.log 2>&1 || echo "FAILED $v $bpc"
  done
  LG_LIB_VARIANT=$v LG_REPS=4 timeout -k 10 300 python tests/diagnostics/bwdnorm_probe.py > gpurun_out/bwdnorm/probe_${v:-product}.log 2>&1 || echo "FAILED probe $v"
done
grep -c "differing elements [1-9]" gpurun_out/bwdnorm/pattern_*.log; grep -h "^weights" gpurun_out/bwdnorm/probe_*.log | sort | uniq -c | sort -rn | head
