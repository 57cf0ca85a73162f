cd $GRAFT_REPO_ROOT
O=gpurun_out/r2w; mkdir -p $O
timeout 1500 python -m pytest tests/test_icp_gpu.py tests/test_golden.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k configs2 > $O/pytest2.log 2>&1; tail -2 $O/pytest2.log
timeout 300 python tools/bench_search.py 2>/dev/null | tail -1 | tee $O/search_vga.json
timeout 300 python tools/bench_search.py --hd 2>/dev/null | tail -1 | tee $O/search_hd.json
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ("frames_per_s","ms_per_frame","icp_iterations_per_frame","ms_model_cloud_frame_cloud_icp_integrate","ms_per_icp","ms_per_iteration","iterations")})'
for rep in 1 2; do
for LIB in $PWD/_ab/libo3d_base.so ""; do
  export O3DMI_LIB=$LIB; echo "=== lib ${LIB:-new} rep $rep"
  echo -n "vga      "; timeout 300 python tools/bench_slam.py --mode slam --vga --frames 60 --no-cpu 2>/dev/null | python -c "$P"
  echo -n "720      "; timeout 300 python tools/bench_slam.py --mode slam --frames 60 --no-cpu 2>/dev/null | python -c "$P"
  echo -n "icp      "; timeout 300 python tools/bench_slam.py --mode icp --no-cpu 2>/dev/null | python -c "$P"
done; done
export O3DMI_LIB=
O3DMI_ICP_TIMING=2 timeout 300 python tools/bench_slam.py --mode slam --vga --frames 30 --no-cpu --phases > $O/t2_phases.log 2>&1
grep "whole call" $O/t2_phases.log | sed -n 8,16p; tail -1 $O/t2_phases.log | cut -c1-500
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vga -o vga -- python $GRAFT_REPO_ROOT/tools/bench_slam.py --mode slam --vga --frames 60 --no-cpu > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_vga -name "*kernel_stats.csv" | head -1); cp "$f" $O/slam_vga_kernel_stats.csv
