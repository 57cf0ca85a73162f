cd $GRAFT_REPO_ROOT
echo "--- self-pinned"
for i in 1 2 3 4 5 6 7 8; do timeout 60 examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | cut -d' ' -f2 | tr '\n' ' '; done; echo
for i in 1 2 3 4 5 6 7 8; do timeout 60 examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | cut -d' ' -f2 | tr '\n' ' '; done; echo
echo "--- O3DMI_EXAMPLE_NO_PIN=1"
for i in 1 2 3 4 5 6 7 8; do O3DMI_EXAMPLE_NO_PIN=1 timeout 60 examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | cut -d' ' -f2 | tr '\n' ' '; done; echo
for i in 1 2 3 4 5 6 7 8; do O3DMI_EXAMPLE_NO_PIN=1 timeout 60 examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*' | cut -d' ' -f2 | tr '\n' ' '; done; echo
timeout 120 examples/icp_slam 12 320 240 0 3 loopback | tail -1
