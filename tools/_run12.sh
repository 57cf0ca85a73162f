cd $GRAFT_REPO_ROOT
O=gpurun_out/r12; mkdir -p $O
timeout 900 python bench.py --no-configs4 --no-pmc --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r12/bench2.json').read().strip().splitlines()[-1])
print(d['value'], d.get('loop_frames_per_s'))
print(json.dumps(d.get('configs2')))
print(len(open('gpurun_out/r12/bench2.json').read().strip().splitlines()[-1]))
PY
tail -3 $O/bench2.err
