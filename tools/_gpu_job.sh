#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5f; mkdir -p $O
timeout -k 5 240 python -m pytest tests/test_raycast_sharded_gpu.py -x -q -m gpu > $O/t_raycast.log 2>&1; tail -2 $O/t_raycast.log
timeout -k 5 300 python -m pytest tests/test_vbg_gpu.py tests/test_vbg_io_gpu.py tests/test_slam_gpu.py -x -q -m gpu > $O/t_vbg.log 2>&1; tail -2 $O/t_vbg.log
for fpl in 12 16 8; do
timeout -k 5 120 python bench.py --no-secondary --no-cpu-baseline --no-pmc --frames-per-launch $fpl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fpl $fpl', round(d['value']), d['roofline']['avg_kernel_ms'])"
done
