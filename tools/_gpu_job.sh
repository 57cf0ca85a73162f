#!/bin/bash
set -u
O=gpurun_out/r5m; mkdir -p $O
timeout -k 5 200 python -m pytest tests/test_icp_gpu.py -x -q -m gpu -k "gated or multiscale or run_to_run or pose_parity or callback or lock_step" > $O/t_gate.log 2>&1; tail -2 $O/t_gate.log | cut -c1-200
for mode in gate nogate; do
 for res in "640 480" "1280 720"; do
  for i in 1 2 3; do
   if [ $mode = nogate ]; then export O3DMI_ICP_NO_GATE=1; else unset O3DMI_ICP_NO_GATE; fi
   timeout -k 5 60 ./examples/icp_slam 60 $res 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', '$res', d['frames_per_s'], d.get('icp_iterations_per_frame'), d.get('max_translation_error_m'))"
  done
 done
done 2>&1 | tee $O/gate_ab.txt
