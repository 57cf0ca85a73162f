#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=$PWD/gpurun_out/r5t; mkdir -p $O
timeout -k 5 60 python -m pytest tests/test_vbg_gpu.py tests/test_configs_gpu.py -x -q -k "unproject" > $O/unproject.log 2>&1; tail -1 $O/unproject.log | cut -c1-200
timeout -k 5 170 python -m pytest tests/test_icp_gpu.py tests/test_slam_gpu.py -x -q > $O/icp.log 2>&1; grep -n "passed\|failed\|error" $O/icp.log | tail -2 | cut -c1-300
for res in "640 480" "1280 720"; do
  for i in 1 2 3 4; do
    for v in new before; do
      if [ $v = before ]; then export LD_LIBRARY_PATH=$PWD/_ab/before; else unset LD_LIBRARY_PATH; fi
      echo -n "$v $res " >> $O/ab.txt
      timeout -k 5 40 examples/icp_slam 60 $res 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['frames_per_s'], d['icp_iterations_per_frame'], d['max_translation_error_m'])" >> $O/ab.txt
    done
  done
done
unset LD_LIBRARY_PATH
cat $O/ab.txt
