#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5n; mkdir -p $O
timeout -k 5 560 python -m pytest tests -x -q -m gpu > $O/gpu_tests_1.log 2>&1; tail -1 $O/gpu_tests_1.log | cut -c1-200
timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
