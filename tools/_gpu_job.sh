cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_configs_gpu.py tests/test_slam_gpu.py -q -m gpu -x -k "device_resident or multiscale or configs2 or cpp or voxel_down or colored or symmetric" 2>&1 | tail -6
for i in 1 2 3 4 5; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*\|"icp_iterations_per_frame": [0-9.]*\|"host_us[^]]*]' | tr '\n' ' '; echo; done
for i in 1 2 3; do examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " 720p"; done
