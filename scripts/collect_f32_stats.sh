#!/bin/bash
# Kernel-trace summaries of the exact-f32 steps on the GPU box (gpurun from the repo root): C3 at f32 and C2.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${LG_ROUND:-r4}
O=gpurun_out/${R}prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/ktf -o c3f -- python3 bench.py --workload c3 --dtype f32 --steps 3 --warmup 5 --no-cpu-baseline --no-graph-leg > $O/ktf.log 2>&1 || exit 1
python scripts/rocpd_stats.py $O/ktf/c3f_results.db 8 $O/${R}_c3_f32_kernel_stats > /dev/null || exit 1
rocprofv3 --kernel-trace --stats -d $O/kt2 -o c2 -- python3 bench.py --workload c2 --steps 5 --warmup 11 --no-cpu-baseline --no-graph-leg > $O/kt2.log 2>&1 || exit 1
python scripts/rocpd_stats.py $O/kt2/c2_results.db 16 $O/${R}_c2_kernel_stats > /dev/null || exit 1
rm -rf $O/ktf/*.db $O/kt2/*.db
head -24 $O/${R}_c3_f32_kernel_stats.md; head -24 $O/${R}_c2_kernel_stats.md
