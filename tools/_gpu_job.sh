#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5h; mkdir -p $O
timeout -k 5 560 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log | cut -c1-200
for res in "640 480" "1280 720"; do
  for i in 1 2 3; do
   timeout -k 5 60 ./examples/icp_slam 60 $res 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vds18', '$res', d['frames_per_s'], d.get('icp_iterations_per_frame'))"
  done
done 2>&1 | tee $O/icp_slam.txt
