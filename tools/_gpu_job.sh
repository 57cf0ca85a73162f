#!/bin/bash
# scratch job of the round (run through gpurun)
set -u
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests/test_vbg_gpu.py tests/test_configs_gpu.py tests/test_icp_gpu.py tests/test_odometry_gpu.py tests/test_slam_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for mode in 1 0; do
 for res in "640 480" "1280 720"; do
  for i in 1 2 3; do
   O3DMI_ICP_ROW_TAIL=$mode ./examples/icp_slam 60 $res 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('row_tail=$mode', '$res', d['frames_per_s'], d.get('icp_iterations_per_frame'))"
  done
 done
done 2>&1 | tee $O/row_tail_ab.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; cp bench_detail.json $O/ 2>/dev/null
