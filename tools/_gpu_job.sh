cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do examples/icp_slam 12 320 240 0 | grep -o '"max_translation_error_m": [0-9.]*\|"icp_iterations_per_frame": [0-9.]*' | tr '\n' ' '; echo; done
echo "3 ranks"
for i in 1 2 3 4 5 6; do examples/icp_slam 12 320 240 0 3 loopback | grep -o '"max_translation_error_m": [0-9.]*\|"icp_iterations_per_frame": [0-9.]*' | tr '\n' ' '; echo; done
