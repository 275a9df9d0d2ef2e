#!/bin/bash
# Ablation of up_p16 (GPU box): LG_P16_DBG bits 16 no shift-sum reads, 32 no product writes, 64 no source loads, 128 no in-tile barriers
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/p16
echo "== product"; python scripts/bench_patch.py | grep -E "conv1 dgrad"
for bits in ${LG_P16_BITS:-16 32 48 64 128 176 240}; do
  LG_EXTRA_FLAGS="-DLG_P16_DBG=$bits" LG_VARIANT_SOURCES=n3_pgemm.hip python -m littlegan_amd.csrc.build --variant up16d$bits > gpurun_out/p16/build_up$bits.log 2>&1
  echo "== LG_P16_DBG=$bits"
  LG_LIB_VARIANT=up16d$bits python scripts/bench_patch.py | grep -E "conv1 dgrad"
done
