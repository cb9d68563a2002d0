#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 120 tools/experiments/pk/pk_rate > gpurun_out/pk_rate.txt 2>&1
cat gpurun_out/pk_rate.txt
