cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5w
timeout -k 10 120 scripts/probe/bin/tile_write_probe 256 > gpurun_out/r5w/tile_write_256.log 2>&1 && timeout -k 10 120 scripts/probe/bin/tile_write_probe 512 > gpurun_out/r5w/tile_write_512.log 2>&1 && cat gpurun_out/r5w/tile_write_256.log gpurun_out/r5w/tile_write_512.log
