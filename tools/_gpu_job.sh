set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_vbg_gpu.py tests/test_golden.py -x -q -m gpu > gpurun_out/r2c/pytest_vbg.log 2>&1; tail -3 gpurun_out/r2c/pytest_vbg.log
run() { tag=$1; shift; extra=""; if [ "$1" = "--depth-only" ]; then extra="--depth-only"; shift; fi; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 $extra > gpurun_out/r2c/bench_$tag.json 2> gpurun_out/r2c/bench_$tag.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r2c/bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["value"]), "fps  kernel_ms", round(d["roofline"]["avg_kernel_ms"]*1e3,2), "us")
PY
}
run v1 O3DMI_STEP_VARIANT=1
run v1_notab O3DMI_STEP_VARIANT=1 O3DMI_NO_PREP_TABLES=1
run v1_nofuse O3DMI_STEP_VARIANT=1 O3DMI_NO_FUSE=1
run v1_depthonly --depth-only O3DMI_STEP_VARIANT=1
run v0_depthonly --depth-only O3DMI_STEP_VARIANT=0
PMC_SETS="sq1" CALIB=0 tools/profile_step_pmc.sh r2c_v1 O3DMI_STEP_VARIANT=1 2>&1 | tail -4
sed -i 's/--no-secondary"/--no-secondary --depth-only"/' tools/profile_step_pmc.sh
PMC_SETS="sq1" CALIB=0 tools/profile_step_pmc.sh r2c_v1_depthonly O3DMI_STEP_VARIANT=1 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && O3DMI_NO_FUSE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2c/trace_nofuse -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1; find $GRAFT_REPO_ROOT/gpurun_out/r2c/trace_nofuse -name "*kernel_stats.csv" | xargs head -8
