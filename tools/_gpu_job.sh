cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 1500 python -m pytest tests/test_vbg_gpu.py -q -m gpu -x > gpurun_out/r4c/vbg.log 2>&1
grep -n "passed\|failed\|Fatal\|Error\|Aborted\|core" gpurun_out/r4c/vbg.log | head
tail -5 gpurun_out/r4c/vbg.log
timeout 600 python -m pytest tests/test_configs_gpu.py -q -m gpu -x -k configs1 2>&1 | tail -3
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_odometry_gpu.py tests/test_slam_gpu.py -q -m gpu -x > gpurun_out/r4c/icp.log 2>&1
tail -3 gpurun_out/r4c/icp.log
