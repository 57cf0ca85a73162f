cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_odometry_gpu.py tests/test_slam_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
for i in 1 2 3; do python tools/bench_slam.py --mode model --vga --no-cpu 2>/dev/null | tail -1 | grep -o '"frames_per_s": [0-9.]*'; done
for i in 1 2; do python tools/bench_slam.py --mode model --hd --no-cpu 2>/dev/null | tail -1 | grep -o '"frames_per_s": [0-9.]*'; done
