cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_icp_gpu.py -q -m gpu -x -k "gated or multiscale or device_resident or in_launch" 2>&1 | tail -5
for i in 1 2 3 4; do timeout 120 examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " vga gated"; O3DMI_ICP_NO_GATE=1 timeout 120 examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " vga plain"; done
for i in 1 2 3; do timeout 120 examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " 720p gated"; O3DMI_ICP_NO_GATE=1 timeout 120 examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " 720p plain"; done
