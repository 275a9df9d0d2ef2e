# Round 5: grid cap of the element-wise InstanceNorm passes (blocks of 256 threads walking the map in trips): 8192 / 2046 (bias-sum form) now,
# against fewer, longer-lived blocks — in isolation (scripts/bench_norm.py) and in the C3 step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
L=gpurun_out/r5n/norm_maxblk.log
: > $L
for v in "LG_X=0" "LG_NORM_MAXBLK=4096" "LG_NORM_MAXBLK=2048" "LG_NORM_MAXBLK=1024" "LG_NORM_MAXBLK_DB=1023" "LG_NORM_MAXBLK_DB=510"; do
  echo "== $v" >> $L
  env $v timeout -k 10 200 python scripts/bench_norm.py 2>&1 | grep -v amdgpu.ids >> $L || exit 1
done
for v in "LG_X=0" "LG_NORM_MAXBLK=2048" "LG_NORM_MAXBLK=4096" "LG_NORM_MAXBLK_DB=1023" "LG_X=0" "LG_NORM_MAXBLK=2048"; do
  echo "== C3 step $v" >> $L
  env $v timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $L || exit 1
done
cat $L
