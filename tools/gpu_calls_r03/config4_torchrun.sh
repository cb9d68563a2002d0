#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload config4 --no-cpu-baseline --no-secondary > gpurun_out/bench_config4_torchrun1.json 2> gpurun_out/bench_config4_torchrun1.err || { tail -20 gpurun_out/bench_config4_torchrun1.err; exit 1; }
cut -c1-300 gpurun_out/bench_config4_torchrun1.json
