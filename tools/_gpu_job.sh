#!/bin/bash
mkdir -p gpurun_out/r4za
timeout 900 python -m pytest tests/test_vbg_gpu.py -q -x -k "unproject" -m gpu -s > gpurun_out/r4za/unproj.log 2>&1
echo "unproj rc=$?" >> gpurun_out/r4za/rc.txt
timeout 900 python -m pytest tests/test_configs_gpu.py -q -x -m gpu -s -k "reproducible" > gpurun_out/r4za/cfg.log 2>&1
echo "cfg rc=$?" >> gpurun_out/r4za/rc.txt
tail -5 gpurun_out/r4za/unproj.log; tail -30 gpurun_out/r4za/cfg.log; cat gpurun_out/r4za/rc.txt
