cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
F=$R/gpurun_out/final
# 1. the whole GPU suite, smoke
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED" | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# 2. the driver's bench line
timeout 900 python bench.py > $F/r3z_bench.json 2> $F/r3z_bench.err; echo "bench rc $?"
# 3. kernel stats of the headline leg under rocprofv3 (same kernel, same steps)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr1 -o tr -- python $R/bench.py --no-pmc --no-secondary --no-cpu-baseline > $F/r3z_trace_bench.json 2>/dev/null
cp $(find /tmp/tr1 -name '*kernel_stats.csv' | head -1) $F/r3z_kernel_stats.csv
python $R/tools/kernel_gaps.py $(find /tmp/tr1 -name '*kernel_trace.csv' | head -1) FrameStepKernel > $F/r3z_kernel_gaps.txt 2>&1
rm -rf /tmp/tr1
# 4. tracking loop timelines
for sz in "640 480 vga" "1280 720 720p"; do set -- $sz
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr2 -o tr -- $R/examples/icp_slam 60 $1 $2 > /dev/null 2>&1
python $R/tools/slam_timeline.py $(find /tmp/tr2 -name '*kernel_trace.csv' | head -1) 59 > $F/r3z_slam_timeline_$3.txt
cp $(find /tmp/tr2 -name '*kernel_stats.csv' | head -1) $F/r3z_icp_slam_$3_kernel_stats.csv
rm -rf /tmp/tr2
done
cd $R
# 5. N = 2 dry run of the distributed branch (two ranks share the GPU, gloo)
timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 2 --warmup 1 --batch 1000 --no-pmc --no-secondary --no-cpu-baseline > $F/r3z_dryrun_n2.json 2> $F/r3z_dryrun_n2.err; echo "dryrun rc $?"
# 6. other loops
python tools/bench_slam.py --mode model --vga --no-cpu 2>/dev/null | tail -1 | cut -c1-400 > $F/r3z_model_vga.json
python tools/bench_slam.py --mode model --hd --no-cpu 2>/dev/null | tail -1 | cut -c1-400 > $F/r3z_model_720p.json
python tools/bench_raycast.py --digest 2>/dev/null | tail -1 > $F/r3z_raycast_vga.json
python tools/bench_raycast.py --digest --hd 2>/dev/null | tail -1 > $F/r3z_raycast_720p.json
head -c 600 $F/r3z_bench.json; echo
cat $F/r3z_slam_timeline_vga.txt | head -16
cat $F/r3z_model_vga.json $F/r3z_model_720p.json $F/r3z_raycast_vga.json $F/r3z_raycast_720p.json
head -c 300 $F/r3z_dryrun_n2.json; echo; tail -3 $F/r3z_dryrun_n2.err
