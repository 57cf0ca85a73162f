cd $GRAFT_REPO_ROOT
for i in 1 2 3; do examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " 720p bucketed"; O3DMI_VDS_SORT=1 examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " 720p sort"; done
for i in 1 2 3; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " vga bucketed"; O3DMI_VDS_SORT=1 examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; echo " vga sort"; done
