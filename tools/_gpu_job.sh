cd $GRAFT_REPO_ROOT
for i in 1 2; do examples/icp_slam 60 640 480 0; examples/icp_slam 60 640 480 1; done
