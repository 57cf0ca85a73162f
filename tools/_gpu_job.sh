cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_configs_gpu.py -k "two_processes or sharded or clone or to_device or odometry or io or normals or slam" > gpurun_out/r2j/pytest2.log 2>&1; tail -4 gpurun_out/r2j/pytest2.log
