cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
for i in 1 2 3; do timeout 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -s -k configs2 2>&1 | grep "configs\[\|passed\|failed\|Error" ; done
timeout 2400 python -m pytest tests/test_configs_gpu.py tests/test_icp_gpu.py tests/test_normals_gpu.py -x -q -m gpu -s > gpurun_out/r2d/pytest.log 2>&1; grep -v "^$" gpurun_out/r2d/pytest.log | grep -i "configs\[\|schedule cloud\|color gradients\|colored multi\|passed\|failed\|Error\|error\|assert" | head -40; tail -3 gpurun_out/r2d/pytest.log
