cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2r/pytest.log 2>&1; tail -3 gpurun_out/r2r/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2r/bench.json 2> gpurun_out/r2r/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2r/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "timed_s", round(d["config"]["timed_region_s"],3))
r=d["roofline"]; print({k:r.get(k) for k in ("frac","frac_hbm","frac_valu","avg_kernel_ms","traffic","valu_insts_per_launch","avg_waves_per_simd","frames_per_launch","traffic_source")})
s=d.get("secondary",{})
for k,v in s.items():
    print(k, {a:v[a] for a in v if a in ("ms_per_icp","ms_per_iteration","frames_per_s","ms_per_frame","cpu_oracle_ms_per_icp","cpu_oracle_ms_per_multiscale_icp","error")}, "frac", v.get("roofline",{}).get("frac") if isinstance(v,dict) else None)
print(d.get("cpu_baseline"))
PY
tail -2 gpurun_out/r2r/bench.err
