#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cylinders.py tests/test_gpu_capsules.py tests/test_gpu_exp05.py -q -m gpu > $O/gpu_suite_g.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite_g.log; tail -15 $O/gpu_suite_g.log | cut -c1-300
bash tools/gpu_calls_r05_f.sh
