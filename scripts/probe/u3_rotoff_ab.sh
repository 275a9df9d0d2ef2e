mkdir -p gpurun_out/r5l
timeout -k 10 400 python -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r5l/tests.log 2>&1; echo "tests rc=$?" 
for ro in 0 2 1 0 2; do
  LG_U3_ROTOFF=$ro timeout -k 10 200 python scripts/bench_gstack.py gpurun_out/r5l/gstack_rot${ro}_$RANDOM.json > gpurun_out/r5l/gstack_rot$ro.log 2>&1
  echo "rotoff $ro:"; grep -i "convT4\|stack\|total" gpurun_out/r5l/gstack_rot$ro.log | tail -3
done
