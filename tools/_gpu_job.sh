cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 2400 python -m pytest tests/test_vbg_gpu.py tests/test_golden.py tests/test_slam_gpu.py tests/test_configs_gpu.py tests/test_icp_gpu.py -x -q -m gpu -k "not stress" > gpurun_out/r2h/pytest.log 2>&1; tail -4 gpurun_out/r2h/pytest.log
for m in "--vga" ""; do python tools/bench_slam.py --mode slam $m --frames 60 --frame-step 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('workload','frames_per_s','ms_per_frame','icp_iterations_per_frame')})"; done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2h/trace_model -o trace -- python $GRAFT_REPO_ROOT/tools/bench_slam.py --mode model --frames 60 > $GRAFT_REPO_ROOT/gpurun_out/r2h/model_vga.json 2>/dev/null; f=$(find $GRAFT_REPO_ROOT/gpurun_out/r2h/trace_model -name "*kernel_stats.csv"); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows:
    n=r["Name"].replace("void o3dmi::(anonymous namespace)::","")[:60]
    if "RayCast" in n or "EstimateRange" in n or "FrameStep" in n or "Odometry" in n or "P2Plane" in n:
        print("%-60s calls %6s avg %9.1f us  pct %s"%(n,r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
cp $f $GRAFT_REPO_ROOT/gpurun_out/r2h/model_vga_kernel_stats.csv; find $GRAFT_REPO_ROOT/gpurun_out/r2h/trace_model -type f -delete
tail -c 600 $GRAFT_REPO_ROOT/gpurun_out/r2h/model_vga.json
