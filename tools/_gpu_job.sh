cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_configs_gpu.py tests/test_slam_gpu.py tests/test_sharding.py -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2 3 4 5; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; done; echo " vga"
for i in 1 2 3 4 5; do examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; done; echo " 720p"
O3DMI_ICP_TIMING=2 examples/icp_slam 12 640 480 2>&1 | grep "whole call" | tail -3
