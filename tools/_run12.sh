cd $GRAFT_REPO_ROOT
O=gpurun_out/r12; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cp bench_detail.json $O/bench_detail.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r12/bench.json').read().strip().splitlines()[-1])
print(d['value'], d.get('loop_frames_per_s'))
print(json.dumps(d.get('configs2')))
print(d.get('drop_in')); print(d['roofline']['frac'], d['roofline']['avg_kernel_ms'])
PY
