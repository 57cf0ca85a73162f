cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_vbg_gpu.py tests/test_slam_gpu.py -q -m gpu -k "raycast or ray_cast or slam or block_coordinates" 2>&1 | grep -E "passed|failed|error" | tail -2
rc() { python tools/bench_raycast.py --digest "$@" 2>/dev/null | tail -1 | cut -c60-300; }
for v in 0 1 0 1; do echo "short div $v"; O3DMI_RAYCAST_SHORT_DIV=$v rc; done
for v in 0 1; do echo "hd short div $v"; O3DMI_RAYCAST_SHORT_DIV=$v rc --hd; done
