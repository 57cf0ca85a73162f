cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final2
F=$R/gpurun_out/final2
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error|FAILED" | tail -5
timeout 600 python bench.py > $F/r3z_bench.json 2> $F/r3z_bench.err; echo "bench rc $?"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr1 -o tr -- python $R/bench.py --no-pmc --no-secondary --no-cpu-baseline > $F/r3z_trace_bench.json 2>/dev/null
cp $(find /tmp/tr1 -name '*kernel_stats.csv' | head -1) $F/r3z_kernel_stats.csv
python $R/tools/kernel_gaps.py $(find /tmp/tr1 -name '*kernel_trace.csv' | head -1) FrameStepKernel > $F/r3z_kernel_gaps.txt 2>&1
head -c 400 $F/r3z_bench.json; echo
grep -i "FrameStepKernel<unsigned short, unsigned short, true" $F/r3z_kernel_stats.csv | head -2
cat $F/r3z_kernel_gaps.txt
