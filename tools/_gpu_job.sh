#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5g; mkdir -p $O
timeout -k 5 200 python -m pytest tests/test_raycast_sharded_gpu.py -x -q -m gpu > $O/t_raycast.log 2>&1; tail -2 $O/t_raycast.log
timeout -k 5 420 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; cp bench_detail.json $O/ 2>/dev/null
export TMPDIR=/tmp; cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/trace -name '*kernel_stats.csv' | head -2; find $O/trace -name '*.db' -delete; find $O/trace -name '*.csv' -size +8M -delete
