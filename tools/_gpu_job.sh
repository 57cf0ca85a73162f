cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vbg_gpu.py tests/test_slam_gpu.py -q -m gpu -k "raycast or ray_cast or slam or block_coordinates" 2>&1 | grep -E "passed|failed|error" | tail -2
O3DMI_RAYCAST_STEPS=1 python tools/bench_raycast.py --repeat 1 2>&1 | grep "o3dmi" | tail -6
python tools/bench_raycast.py --digest 2>/dev/null | tail -1
