#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5o; mkdir -p $O
timeout -k 5 560 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -1 $O/gpu_tests.log | cut -c1-200
timeout -k 5 420 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; cp bench_detail.json $O/ 2>/dev/null
