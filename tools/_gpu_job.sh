cd $GRAFT_REPO_ROOT
O=gpurun_out/r2y; mkdir -p $O
timeout 1500 python -m pytest tests/test_icp_gpu.py tests/test_normals_gpu.py tests/test_golden.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k configs2 2>&1 | tail -1
timeout 300 python tools/bench_search.py 2>/dev/null | tail -1 | tee $O/search_vga.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for l in d['levels']: print(l['voxel'], l['source_points'], l['search_us'])"
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ("frames_per_s","ms_per_frame","icp_iterations_per_frame","ms_per_icp","ms_per_iteration","iterations")})'
for i in 1 2; do timeout 300 python tools/bench_slam.py --mode icp --no-cpu 2>/dev/null | python -c "$P"; done
for i in 1 2 3; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*\|"host_us[^]]*\]' | tr '\n' ' '; echo; done
examples/icp_slam 60 1280 720 | grep -o '"frames_per_s": [0-9.]*'
