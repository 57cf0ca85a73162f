cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "base: "; LD_LIBRARY_PATH=$PWD/_ab/lib examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*\|"host_us[^]]*\]'| tr '\n' ' '; echo
  echo -n "new : "; examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*\|"host_us[^]]*\]' | tr '\n' ' '; echo
done
echo base; LD_LIBRARY_PATH=$PWD/_ab/lib O3DMI_ICP_TIMING=2 examples/icp_slam 30 640 480 2>&1 | grep "whole call" | sed -n 10,14p
echo new; O3DMI_ICP_TIMING=2 examples/icp_slam 30 640 480 2>&1 | grep "whole call" | sed -n 10,14p
timeout 600 python -m pytest tests/test_icp_gpu.py -x -q -m gpu -k "multiscale or icp_pose or two_ranks or colored" 2>&1 | tail -2
