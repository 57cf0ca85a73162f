cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_icp_gpu.py -x -q -m gpu -k "voxel_down or multiscale or colored or symmetric" 2>&1 | tail -1
for i in 1 2 3 4; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*'; done
O3DMI_ICP_TIMING=2 examples/icp_slam 30 640 480 2>&1 | grep "whole call" | sed -n 12,16p
