cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
export TMPDIR=/tmp
B="python $PWD/bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 3 --warmup 1"
export O3DMI_SUMMARIZE_KERNEL=ChunkIntegrateKernel
for w in 8 1; do
  if [ $w = 1 ]; then ARGS="--force-sliced"; else ARGS="--emulate-world $w"; fi
  rm -rf /tmp/pm_$w
  (cd /tmp && O3DMI_SLICED_RAW=1 timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm_$w -o p -- $B $ARGS > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r4o/pm_$w.err)
  python tools/c4_summarize.py pmc /tmp/pm_$w > gpurun_out/r4o/pmc_chunk_w$w.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r4o/pmc_chunk_w$w.json'))
p=d['per_launch']; cyc=p['GRBM_GUI_ACTIVE']/8
print('w=$w launches',d['launches'],'kernel us',round(d.get('kernel_us_avg_this_pass',0),1),'valu frac',round(p['SQ_ACTIVE_INST_VALU']*4/(1024*cyc),3),'waves/simd',round(p['SQ_WAVE_CYCLES']*4/(1024*cyc),2),'insts',round(p['SQ_INSTS_VALU']/1e6,1),'M; split issue',round(p['SQ_ACTIVE_INST_ANY']/p['SQ_WAVE_CYCLES'],3),'wait',round(p['SQ_WAIT_ANY']/p['SQ_WAVE_CYCLES'],3),'stall',round(p['SQ_WAIT_INST_ANY']/p['SQ_WAVE_CYCLES'],3), 'waves', p['SQ_WAVES'])
PY
done
