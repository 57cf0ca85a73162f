cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
export TMPDIR=/tmp
B="python $PWD/bench.py --no-secondary --no-cpu-baseline"
for d in 0 1 0 1; do
O3DMI_STEP_DEAL=$d timeout 600 $B > gpurun_out/r4m/deal$d.json 2> gpurun_out/r4m/deal$d.err
python -c "
import json;d=json.load(open('gpurun_out/r4m/deal$d.json'));r=d['roofline'];print('deal $d', round(d['value']), r['avg_kernel_ms'], 'read_overfetch', r['read_overfetch'], 'frac_hbm', r['frac_hbm'], 'valu', r['frac_valu'], 'traffic', r['traffic'])"
done
timeout 600 python -m pytest tests/test_vbg_gpu.py -q -m gpu -x -k "frame_batch or fused or long_run" 2>&1 | tail -2
