#!/bin/bash
# Non-temporal accesses in the bf16 norm passes (GPU box): LG_NORM_NT bits 1 backward-apply loads, 2 backward-apply stores, 4 apply loads, 8 apply stores
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/nt
echo "== product"; python scripts/bench_norm.py | grep -v "torch bf16"
for bits in ${LG_NT_BITS:-1 3 4 12 15}; do
  LG_EXTRA_FLAGS="-DLG_NORM_NT=$bits" LG_VARIANT_SOURCES=norm.hip python -m littlegan_amd.csrc.build --variant nnt$bits > gpurun_out/nt/build_$bits.log 2>&1
  echo "== LG_NORM_NT=$bits"
  LG_LIB_VARIANT=nnt$bits python scripts/bench_norm.py | grep -v "torch bf16"
done
