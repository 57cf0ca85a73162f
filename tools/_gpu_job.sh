cd $GRAFT_REPO_ROOT
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(round(d["value"]), d["ms_per_step"], r.get("avg_kernel_ms"), r.get("frac"))'
for rep in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary 2>/dev/null | python -c "$P"; done
timeout 1500 python -m pytest tests/test_configs_gpu.py tests/test_vbg_gpu.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
