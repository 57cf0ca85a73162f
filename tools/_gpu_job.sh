cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/ab_gn.py 2>&1 | tail -40
timeout 600 python -m pytest tests/test_configs_gpu.py -q -m gpu -s -k "frame_sharded_8_ranks" 2>&1 | grep -v "^$" | tail -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fast -o fast --output-format csv -- $GRAFT_REPO_ROOT/examples/icp_slam 60 640 480 > /dev/null 2>&1
O3DMI_ICP_HOST_SOLVE=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_slow -o slow --output-format csv -- $GRAFT_REPO_ROOT/examples/icp_slam 60 640 480 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_fast gpurun_out/prof_slow -name "*kernel_stats.csv" | head
for f in $(find gpurun_out/prof_fast gpurun_out/prof_slow -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-160; done
