cd $GRAFT_REPO_ROOT
R=${LG_ROUND:-r5}
export LG_ROUND=$R
bash scripts/collect_profiles.sh > gpurun_out/${R}_collect.log 2>&1 || { tail -20 gpurun_out/${R}_collect.log; exit 1; }
cp gpurun_out/${R}prof/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json   # bench.py reads the traffic of its dominant kernel from here
python bench.py > gpurun_out/${R}_bench_c3.json 2> gpurun_out/${R}_bench_c3.err || exit 1
python bench.py --dp-contention 8,16,32,64 --no-cpu-baseline > gpurun_out/${R}_bench_c3_contention.json 2> gpurun_out/${R}_bench_c3_contention.err || exit 1
python bench.py --workload c3 --dtype f32 --no-cpu-baseline > gpurun_out/${R}_bench_c3_f32.json 2> gpurun_out/${R}_bench_c3_f32.err || exit 1
python bench.py --workload c2 --cpu-baseline-full > gpurun_out/${R}_bench_c2.json 2> gpurun_out/${R}_bench_c2.err || exit 1
python bench.py --workload c5 --steps 20 --warmup 12 --no-cpu-baseline > gpurun_out/${R}_bench_c5.json 2> gpurun_out/${R}_bench_c5.err || exit 1
for f in c3 c3_f32 c2 c5; do python -c "
import json; d=json.load(open('gpurun_out/'+'${R}'+'_bench_$f.json')); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d.get('graph_replay'))"; done
python -c "
import json; d=json.load(open('gpurun_out/${R}prof/${R}_gstack_forward.json')); print([(l['layer'], l['kernel'], l['us_median']) for l in d['layers']], d['total_us_median'], d['frac_of_bf16_peak'])"
python __graft_entry__.py --smoke 2>&1 | tail -3
