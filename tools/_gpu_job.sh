cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_vbg_gpu.py tests/test_golden.py tests/test_slam_gpu.py -q -m gpu -x -k "raycast or ray_cast or golden or model or last_frame" 2>&1 | tail -3
python tools/bench_raycast.py
O3DMI_RAYCAST_NO_XCD=1 python tools/bench_raycast.py
python tools/bench_raycast.py --hd
O3DMI_RAYCAST_NO_XCD=1 python tools/bench_raycast.py --hd
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rcA -o rc -- python $GRAFT_REPO_ROOT/tools/bench_raycast.py --repeat 20 > /dev/null 2>&1
O3DMI_RAYCAST_NO_XCD=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rcB -o rc -- python $GRAFT_REPO_ROOT/tools/bench_raycast.py --repeat 20 > /dev/null 2>&1
grep RayCastKernel /tmp/rcA/rc_kernel_stats.csv | cut -d, -f2-4 | tail -1
grep RayCastKernel /tmp/rcB/rc_kernel_stats.csv | cut -d, -f2-4 | tail -1
python - <<'PY'
import csv
for d in ("/tmp/rcA","/tmp/rcB"):
    for r in csv.DictReader(open(d+"/rc_kernel_stats.csv")):
        if "RayCastKernel" in r["Name"]: print(d, r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
