# Round 5: per-kernel time of the C5 per-GPU share (256 x 256, B = 256) — does any launch fall back to an older kernel at that geometry?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5c5/kt -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 3 --warmup 11 --no-cpu-baseline --no-graph-leg > $GRAFT_REPO_ROOT/gpurun_out/r5c5/kt.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/r5c5/kt.log; exit 1; }
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/r5c5/kt -name "*.db" | head -1)
python3 scripts/rocpd_stats.py $db 14 gpurun_out/r5c5/r5_c5_kernel_stats > /dev/null
head -40 gpurun_out/r5c5/r5_c5_kernel_stats.md
rm -rf gpurun_out/r5c5/kt
