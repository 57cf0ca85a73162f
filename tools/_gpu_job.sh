cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
B="python $PWD/bench.py --no-secondary --no-cpu-baseline --no-pmc"
for i in 1 2; do
timeout 600 $B > gpurun_out/r4q/n1_$i.json 2> gpurun_out/r4q/n1.err
python -c "
import json;d=json.load(open('gpurun_out/r4q/n1_$i.json'));print('headline', round(d['value']), 'cold', round(d['cold_pass_frames_per_s']), d['roofline']['avg_kernel_ms'])"
O3DMI_STRICT_CAPACITY=1 timeout 600 $B --block-count 524288 > gpurun_out/r4q/n1_strict_$i.json 2> gpurun_out/r4q/n1.err
python -c "
import json;d=json.load(open('gpurun_out/r4q/n1_strict_$i.json'));print('strict 524288', round(d['value']), 'cold', round(d['cold_pass_frames_per_s']), d['roofline']['avg_kernel_ms'])"
done
timeout 600 python -m pytest tests/test_vbg_gpu.py -q -m gpu -x -k "overflow or run_ahead or frame_batch" 2>&1 | tail -2
