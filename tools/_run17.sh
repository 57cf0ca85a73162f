cd $GRAFT_REPO_ROOT
O=gpurun_out/r17; mkdir -p $O
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
cp bench_detail.json $O/bench_detail.json 2>/dev/null
python - <<'PY'
import json
l=open('gpurun_out/r17/bench.json').read().strip().splitlines()[-1]
d=json.loads(l)
print(len(l), d['value'], d.get('loop_frames_per_s'), d.get('loop_frames_per_s_with_first_frame'))
print(json.dumps(d.get('configs2')))
PY
