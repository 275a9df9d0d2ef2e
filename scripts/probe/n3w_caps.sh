#!/bin/bash
# conv1 / final weight gradient against the persistent grid size (GPU box): LG_N3W_CAP64 / LG_N3W_CAP32 sweeps; LB4=1 adds the variant compiled for four blocks per CU
set -e
cd "$(dirname "$0")/../.."
for c in 512 768 1024; do echo "== CAP64=$c"; LG_N3W_CAP64=$c python scripts/probe/n3w_ab.py | grep conv1; done
if [ -n "$LB4" ]; then
  LG_EXTRA_FLAGS="-DLG_N3W_LB=4" LG_VARIANT_SOURCES=n3_kernels.hip python -m littlegan_amd.csrc.build --variant n3wlb4 > /dev/null 2>&1
  for c in 768 1024; do echo "== LB4 CAP64=$c"; LG_LIB_VARIANT=n3wlb4 LG_N3W_CAP64=$c python scripts/probe/n3w_ab.py | grep conv1; done
fi
for c in 512 768 1024 1536; do echo "== CAP32=$c"; LG_N3W_CAP32=$c python scripts/probe/n3w_ab.py | grep final; done
