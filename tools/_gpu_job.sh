#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5i; mkdir -p $O
timeout -k 5 240 python -m pytest tests/test_vbg_gpu.py tests/test_configs_gpu.py tests/test_raycast_sharded_gpu.py -x -q -m gpu -k "not sliced and not slice" > $O/tests.log 2>&1; tail -2 $O/tests.log | cut -c1-200
for i in 1 2; do
timeout -k 5 100 python bench.py --no-secondary --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer', round(d['value']), d['roofline']['avg_kernel_ms'])"
O3DMI_LIB=$PWD/_ab/nodefer/libo3d_mi355x.so timeout -k 5 100 python bench.py --no-secondary --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nodefer', round(d['value']), d['roofline']['avg_kernel_ms'])"
done
