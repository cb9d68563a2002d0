#!/bin/bash
set -e
mkdir -p gpurun_out
python tools/experiments/load_order.py 65536 300 > gpurun_out/load_order.txt 2>&1 || { tail -30 gpurun_out/load_order.txt; exit 1; }
python tools/experiments/load_order.py 32768 300 >> gpurun_out/load_order.txt 2>&1
cat gpurun_out/load_order.txt
