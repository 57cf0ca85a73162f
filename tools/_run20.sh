cd $GRAFT_REPO_ROOT
O=gpurun_out/r20; mkdir -p $O
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
cp bench_detail.json $O/bench_detail.json 2>/dev/null
python - <<'PY'
import json
l=open('gpurun_out/r20/bench.json').read().strip().splitlines()[-1]
d=json.loads(l)
print(len(l), d['value'], d.get('loop_frames_per_s'), d.get('loop_frames_per_s_with_first_frame'))
print(json.dumps(d.get('configs2')))
PY
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
