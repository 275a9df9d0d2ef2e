#!/bin/bash
# Ablation of n3_wgrad16 (GPU box): variant builds of n3_kernels.hip with LG_N3W_DBG bits (results wrong, timing only), timed by n3w_ab.py.
#   1 no plane scatter, 2 no 3-channel loads, 4 no wide-operand loads, 8 no MFMA, 16 no wide-operand LDS stores
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/n3w
for bits in ${LG_N3W_BITS:-0 1 2 4 6 8 16 31}; do
  LG_EXTRA_FLAGS="-DLG_N3W_DBG=$bits" LG_VARIANT_SOURCES=n3_kernels.hip python -m littlegan_amd.csrc.build --variant n3w$bits > gpurun_out/n3w/build_$bits.log 2>&1
  echo "== LG_N3W_DBG=$bits"
  LG_LIB_VARIANT=n3w$bits python scripts/probe/n3w_ab.py
done
