#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=$PWD/gpurun_out/r5s; mkdir -p $O
timeout -k 5 120 python tools/bench_api.py 300 > $O/api.json 2> $O/api.err; cat $O/api.json | cut -c1-900; tail -2 $O/api.err
timeout -k 5 520 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; grep -n "passed\|failed\|error" $O/gpu_tests.log | tail -3 | cut -c1-300
