cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
export TMPDIR=/tmp
B="python $PWD/bench.py --no-secondary --no-cpu-baseline --no-pmc"
for w in 8; do
timeout 600 $B --emulate-world $w > gpurun_out/r4g/emu_w$w.json 2> gpurun_out/r4g/emu_w$w.err
python -c "
import json;d=json.load(open('gpurun_out/r4g/emu_w$w.json'));print('emu $w', d['value'], d['roofline']['avg_kernel_ms'], d['config']['active_blocks'])"
done
timeout 600 $B --force-sliced > gpurun_out/r4g/n1_sliced.json 2> gpurun_out/r4g/n1_sliced.err
python -c "
import json;d=json.load(open('gpurun_out/r4g/n1_sliced.json'));print('n1 sliced', d['value'], d['roofline']['avg_kernel_ms'], d['config']['active_blocks'])"
/usr/bin/time -v timeout 1500 python bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err
tail -25 gpurun_out/r4g/bench.err | grep -E "Elapsed|Maximum resident|Error|error" 
wc -c gpurun_out/r4g/bench.json
cat gpurun_out/r4g/bench.json
cp bench_detail.json gpurun_out/r4g/bench_detail.json 2>/dev/null
