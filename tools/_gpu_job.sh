cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for fpl in 12 16; do
python bench.py --no-secondary --no-cpu-baseline --no-pmc --frames-per-launch $fpl --block-count 524288 > gpurun_out/r3l_bench_fpl$fpl.json 2> gpurun_out/r3l_bench_fpl$fpl.err
done
python bench.py --no-secondary --no-cpu-baseline --no-pmc --block-count 524288 > gpurun_out/r3l_bench_fpl8_big.json 2> /dev/null
python - <<'PY'
import json
for f in ("fpl12","fpl16","fpl8_big"):
    d=json.loads(open("gpurun_out/r3l_bench_%s.json"%f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "value %.0f cold %.0f kms %.4f frac %.3f equiv %.3f"%(d["value"], d["cold_pass_frames_per_s"], r["avg_kernel_ms"], r["frac"], r["equivalent_frac"]))
PY
