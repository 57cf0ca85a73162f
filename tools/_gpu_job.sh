cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_icp_gpu.py -q -m gpu -k "carries_the_next" 2>&1 | tail -15
