cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_vbg_gpu.py tests/test_golden.py tests/test_slam_gpu.py -x -q -m gpu 2>&1 | tail -3
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(round(d["value"]), d["ms_per_step"], r.get("avg_kernel_ms"), r.get("frames_per_launch"), r.get("frac"))'
for rep in 1 2; do
echo -n "fpl 8  cap 262144: "; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary 2>/dev/null | python -c "$P"
echo -n "fpl 16 cap 524288: "; timeout 300 python bench.py --steps 10 --warmup 3 --frames-per-launch 16 --block-count 524288 --no-cpu-baseline --no-pmc --no-secondary 2>/dev/null | python -c "$P"
echo -n "fpl 12 cap 524288: "; timeout 300 python bench.py --steps 10 --warmup 3 --frames-per-launch 12 --block-count 524288 --no-cpu-baseline --no-pmc --no-secondary 2>/dev/null | python -c "$P"
done
