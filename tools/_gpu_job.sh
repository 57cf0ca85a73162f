#!/bin/bash
set -u
O=gpurun_out/r5l; mkdir -p $O
for v in fine uncached host; do
  for w in 64 256 512; do
    timeout -k 2 30 ./tools/probe_gate $v $w 2>&1 | tail -2
  done
done | tee $O/probe_gate_sfence.txt
