#!/bin/bash
# Ablation of patch_p16 (GPU box): variant builds of n3_pgemm.hip with LG_P16_DBG bits (results wrong, timing only), timed by bench_patch.py.
#   1 no image loads, 2 no output stores, 4 no MFMA, 8 no moments
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/p16
echo "== product"; python scripts/bench_patch.py | grep -E "conv1 fwd|final dgrad"
for bits in ${LG_P16_BITS:-1 2 4 8 15}; do
  LG_EXTRA_FLAGS="-DLG_P16_DBG=$bits" LG_VARIANT_SOURCES=n3_pgemm.hip python -m littlegan_amd.csrc.build --variant p16d$bits > gpurun_out/p16/build_$bits.log 2>&1
  echo "== LG_P16_DBG=$bits"
  LG_LIB_VARIANT=p16d$bits python scripts/bench_patch.py | grep -E "conv1 fwd|final dgrad"
done
for gsz in 512 768 1024 1536; do echo "== LG_PATCH_GRID=$gsz"; LG_PATCH_GRID=$gsz python scripts/bench_patch.py | grep -E "conv1 fwd|final dgrad"; done
