#!/bin/bash
# scratch job of the round (run through gpurun); every step under its own timeout
set -u
O=gpurun_out/r5j; mkdir -p $O
timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --dist-backend gloo > $O/dry_n2.json 2> $O/dry_n2.err; tail -c 700 $O/dry_n2.json; tail -3 $O/dry_n2.err | cut -c1-300
timeout -k 5 200 python bench.py --gpus 2 --steps 2 --warmup 1 --dist-backend gloo > $O/dry_spawn_n2.json 2> $O/dry_spawn_n2.err; tail -c 300 $O/dry_spawn_n2.json
