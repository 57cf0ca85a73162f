cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_vbg_gpu.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
O3DMI_STEP_VARIANT=3 timeout 900 python -m pytest tests/test_vbg_gpu.py -x -q -m gpu -k "frame or stream or group" 2>&1 | tail -1
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(round(d["value"]), d["ms_per_step"], r.get("avg_kernel_ms"), r.get("frac"))'
for rep in 1 2; do
for V in 2 3; do
  echo -n "variant $V: "; O3DMI_STEP_VARIANT=$V timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary 2>/dev/null | python -c "$P"
done; done
