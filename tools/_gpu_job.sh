#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5p; mkdir -p $O
timeout -k 5 150 python -m pytest tests/test_raycast_sharded_gpu.py -x -q > $O/sharded.log 2>&1; tail -3 $O/sharded.log | cut -c1-300
timeout -k 5 420 python -m pytest tests -x -q -m gpu --deselect tests/test_raycast_sharded_gpu.py > $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log | cut -c1-200
