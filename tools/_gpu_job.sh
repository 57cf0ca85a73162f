#!/bin/bash
O=gpurun_out/r4zi; mkdir -p $O
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-configs4"
for w in 2 4 8; do
 for raw in 0 1; do
  O3DMI_SLICED_RAW=$raw timeout 100 $B --emulate-world $w > $O/emu_w${w}_raw${raw}.json 2> $O/err.txt
 done
done
O3DMI_CHUNK_TIMELINE=$O/tl.bin O3DMI_CHUNK_TIMELINE_LAUNCH=200 timeout 100 $B --steps 6 --emulate-world 8 > $O/emu_w8_default_tl.json 2>> $O/err.txt
python tools/chunk_timeline.py $O/tl.bin --json > $O/chunk_timeline_w8_final.json; rm -f $O/tl.bin
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4zi/emu*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), d['roofline'].get('avg_kernel_ms'))
    except Exception as e:
        print(f, 'ERR', e)
d=json.load(open('gpurun_out/r4zi/chunk_timeline_w8_final.json'))
print(d['span_us'], d['cu_end_us'], d['resident_workgroups_at_5pct_steps'])
PY
