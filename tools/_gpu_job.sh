cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_search.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for l in d['levels']: print(l['voxel'], l['source_points'], l['search_us'])"
for i in 1 2 3; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*'; done
timeout 600 python -m pytest tests/test_icp_gpu.py -x -q -m gpu -k "pose or multiscale or sums or hybrid" 2>&1 | tail -1
