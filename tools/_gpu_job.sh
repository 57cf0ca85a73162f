cd $GRAFT_REPO_ROOT
O=gpurun_out/r2z; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2z/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "timed_s", round(d["config"]["timed_region_s"],3), "ms_per_step", d["ms_per_step"])
r=d["roofline"]; print({k:r.get(k) for k in ("frac","frac_hbm","frac_valu","avg_kernel_ms","wall_ms_per_launch","traffic","avg_waves_per_simd","traffic_source")})
s=d.get("secondary",{})
for k,v in s.items():
    if isinstance(v,dict): print(k, {a:v[a] for a in v if a in ("ms_per_icp","ms_per_iteration","frames_per_s","ms_per_frame","icp_iterations_per_frame","cpu_oracle_ms_per_icp","cpu_oracle_ms_per_multiscale_icp","error","frames_per_s_of_5_runs")}, "frac", v.get("roofline",{}).get("frac"))
print(d.get("cpu_baseline"))
PY
tail -2 $O/bench.err
