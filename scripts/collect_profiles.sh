#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel trace (+ per-kernel summary), HBM-side traffic (FETCH_SIZE / WRITE_SIZE in separate passes), MFMA busy, wave states.
# Counter passes carry --pmc only (no trace domains), as the pool requires.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${LG_ROUND:-r5}
O=gpurun_out/${R}prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o c3 -- python3 bench.py --steps 5 --warmup 11 --no-cpu-baseline --no-graph-leg > $O/kt.log 2>&1 || exit 1
python scripts/rocpd_stats.py $O/kt/c3_results.db 16 $O/${R}_c3_kernel_stats > /dev/null || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o f -- python3 bench.py --steps 2 --warmup 11 --no-cpu-baseline --no-graph-leg > $O/pf.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o w -- python3 bench.py --steps 2 --warmup 11 --no-cpu-baseline --no-graph-leg > $O/pw.log 2>&1 || exit 1
python scripts/pmc_traffic.py $(ls $O/pf/*counter_collection.csv | head -1) $(ls $O/pw/*counter_collection.csv | head -1) $O/${R}_pmc_traffic.json > /dev/null || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pm -o m -- python3 bench.py --steps 2 --warmup 11 --no-cpu-baseline --no-graph-leg > $O/pm.log 2>&1 || exit 1
python scripts/pmc_mfma_util.py $(ls $O/pm/*counter_collection.csv | head -1) $O/${R}_c3_mfma_util.md > /dev/null || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/ps -o s -- python3 bench.py --steps 2 --warmup 11 --no-cpu-baseline --no-graph-leg > $O/ps.log 2>&1 || exit 1
python scripts/pmc_sq_waits.py $(ls $O/ps/*counter_collection.csv | head -1) $O/${R}_c3_sq_waits.md > /dev/null || exit 1
rm -rf $O/pf $O/pw $O/pm $O/ps $O/kt/*.db
python3 scripts/bench_gstack.py $O/${R}_gstack_forward.json > $O/gstack.log 2>&1 || exit 1
# the north star's literal metric: MFMA busy over the FORWARD launches of the Generator's transposed-conv stack only (bench_gstack.py launches
# nothing else but the untimed apply passes and the packs, which the summary leaves out), time-weighted, final layer included
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pg -o g -- python3 scripts/bench_gstack.py $O/gstack_pmc_pass.json > $O/pg.log 2>&1 || exit 1
python scripts/pmc_mfma_util.py $(ls $O/pg/*counter_collection.csv | head -1) $O/${R}_gstack_mfma_busy.md --stack > /dev/null || exit 1
rm -rf $O/pg $O/gstack_pmc_pass.json
ls -la $O
