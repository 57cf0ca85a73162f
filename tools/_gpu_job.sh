cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
timeout 1800 python -m pytest tests/test_vbg_gpu.py tests/test_golden.py tests/test_slam_gpu.py tests/test_configs_gpu.py tests/test_vbg_io_gpu.py -x -q -m gpu > gpurun_out/r2o/pytest.log 2>&1; tail -3 gpurun_out/r2o/pytest.log
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 10 --warmup 3 > gpurun_out/r2o/bench_$tag.json 2> gpurun_out/r2o/bench_$tag.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r2o/bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["value"]), "fps  kernel", round(d["roofline"]["avg_kernel_ms"]*1e3,2), "us")
PY
}
run form1 O3DMI_STEP_VARIANT=1
run form0 O3DMI_STEP_VARIANT=0
