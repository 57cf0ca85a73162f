#!/bin/bash
O=gpurun_out/r4zk; mkdir -p $O
T=tests/test_vbg_gpu.py::test_sliced_touch_ownership_union_is_the_single_grid
timeout 35 python -m pytest -q -x -m gpu "$T[1-4-False-True-True-False]" "$T[3-2-False-True-True-False]" "$T[8-3-False-True-True-False]" "$T[8-12-False-True-True-False]" "$T[2-16-True-True-True-False]" "$T[4-5-False-False-True-False]" "$T[8-12-False-True-True-True]" > $O/raw_tests.log 2>&1
echo "tests rc=$?" > $O/rc.txt
tail -2 $O/raw_tests.log; cat $O/rc.txt
