cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_icp_gpu.py tests/test_configs_gpu.py -q -m gpu -x -k "voxel_down or multiscale or configs2 or colored or level_sharded" 2>&1 | tail -2
for i in 1 2 3 4 5; do examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; done; echo " 720p"
for i in 1 2 3; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; done; echo " vga"
