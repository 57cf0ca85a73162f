cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vbg_gpu.py -q -m gpu -k "partial_tiles or long_crawls" 2>&1 | grep -v "^$" | tail -30
O3DMI_RAYCAST_COOP=0 timeout 900 python -m pytest tests/test_vbg_gpu.py -q -m gpu -k "partial_tiles or long_crawls" 2>&1 | tail -3
