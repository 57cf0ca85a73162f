cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED" | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r3z_bench.json 2> gpurun_out/r3z_bench.err; echo "bench rc $?"
tail -c 1500 gpurun_out/r3z_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3z_bench.json'))
print({k:d[k] for k in d if k not in ('secondary','roofline','cpu_baseline','config')})
print(d['roofline'])
s=d['secondary']
for k,v in s.items():
    if isinstance(v,dict):
        print(k, {kk:v[kk] for kk in v if kk in ('frames_per_s','frames_per_s_of_5_runs','ms_per_icp','value','ms_per_frame')})
PY
