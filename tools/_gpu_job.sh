#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5q; mkdir -p $O
timeout -k 5 200 python -m pytest tests/test_raycast_sharded_gpu.py -x -q > $O/sharded.log 2>&1; tail -3 $O/sharded.log | cut -c1-300
timeout -k 5 240 python -m pytest tests/test_slam_gpu.py -x -q > $O/slam.log 2>&1; tail -3 $O/slam.log | cut -c1-300
timeout -k 5 60 examples/icp_slam 12 320 240 0 3 loopback > $O/icp_slam_3ranks.json 2>&1; cat $O/icp_slam_3ranks.json | cut -c1-400
timeout -k 5 60 examples/icp_slam 12 320 240 0 > $O/icp_slam_1rank.json 2>&1; tail -1 $O/icp_slam_1rank.json | cut -c1-300
