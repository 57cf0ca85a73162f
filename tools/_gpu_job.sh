cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_icp_gpu.py -q -m gpu -x -k "voxel_down" 2>&1 | tail -2
for i in 1 2 3 4; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo -n " bucketed | "; O3DMI_VDS_SORT=1 examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " sort"; done
for i in 1 2 3; do examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; done; echo " 720p"
