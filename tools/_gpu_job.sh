#!/bin/bash
# scratch job of the round (run through gpurun)
set -u
O=gpurun_out/r5d; mkdir -p $O
python -m pytest tests/test_icp_gpu.py tests/test_slam_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
python -m pytest tests/test_vbg_gpu.py -x -q -m gpu -k "slice or shard or sliced" > $O/tests_sliced.log 2>&1; tail -2 $O/tests_sliced.log
for res in "640 480" "1280 720"; do
  for i in 1 2 3 4 5; do
   ./examples/icp_slam 60 $res 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sealed_tail', '$res', d['frames_per_s'], d.get('icp_iterations_per_frame'))"
  done
done 2>&1 | tee $O/icp_slam.txt
bash tools/emu_table.sh r5d "2 4 8" "0 1" 2>&1 | tail -20
