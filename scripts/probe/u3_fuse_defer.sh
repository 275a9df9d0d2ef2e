# Round 5: conv_up3's FUSE form (data gradient + norm-backward sums) with the deferred row sweep — parity first, then timing
mkdir -p gpurun_out/r5s
timeout -k 10 600 python -m pytest tests/test_launch_shapes_gpu.py tests/test_ops_gpu.py tests/test_step_replay_gpu.py -m gpu -x -q -k "fused or persistent or replay or dgrad or data_gradient or whole" > gpurun_out/r5s/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5s/tests.log
for b in 256 512; do echo "B=$b"; LG_B=$b LG_FUSE=1 timeout -k 10 120 python scripts/bench_layer.py "conv2 dgrad" 2>&1 | grep -v amdgpu; LG_B=$b LG_FUSE=0 timeout -k 10 120 python scripts/bench_layer.py "conv2 dgrad" "convT3 fwd" 2>&1 | grep -v amdgpu; done
