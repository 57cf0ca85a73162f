cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_vbg_gpu.py tests/test_slam_gpu.py tests/test_abi.py -x -q -m gpu -k "last_frame or cpp or raycast or ray_cast" 2>&1 | tail -3
for i in 1 2 3; do examples/icp_slam 60 640 480 0; examples/icp_slam 60 640 480 1; done
for i in 1 2; do examples/icp_slam 60 1280 720 0; examples/icp_slam 60 1280 720 1; done
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ("frames_per_s","ms_per_frame","icp_iterations_per_frame","final_pose_err_rad_m")})'
for i in 1 2; do timeout 300 python tools/bench_slam.py --mode slam --vga --frames 60 --no-cpu 2>/dev/null | python -c "$P"; done
timeout 300 python tools/bench_slam.py --mode slam --frames 60 --no-cpu 2>/dev/null | python -c "$P"
