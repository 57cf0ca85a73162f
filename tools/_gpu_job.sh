cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vbg_gpu.py -q -m gpu -x -k "sliced_touch_ownership" > gpurun_out/r4p/vbg.log 2>&1
grep -n "passed\|failed\|Fatal\|Aborted" gpurun_out/r4p/vbg.log | head -3
B="python $PWD/bench.py --no-secondary --no-cpu-baseline --no-pmc"
for w in 8 4; do
 for raw in 1 0; do
  for pipe in 0 1; do
   O3DMI_SLICED_RAW=$raw O3DMI_SLICED_PIPE=$pipe timeout 600 $B --emulate-world $w > gpurun_out/r4p/emu_w${w}_raw${raw}_pipe${pipe}.json 2> gpurun_out/r4p/emu.err
   python -c "
import json;d=json.load(open('gpurun_out/r4p/emu_w${w}_raw${raw}_pipe${pipe}.json'));print('emu w$w raw$raw pipe$pipe', round(d['value']), d['roofline']['avg_kernel_ms'])"
  done
 done
done
