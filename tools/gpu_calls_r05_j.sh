#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cylinders.py tests/test_gpu_capsules.py tests/test_gpu_random_robots.py -q -m gpu > $O/gpu_suite_j.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite_j.log; tail -8 $O/gpu_suite_j.log | cut -c1-300
timeout -k 10 400 python tools/fuzz_parity.py --seeds 510000 513000 --minutes 3 --log $O/fuzz_j.log > $O/fuzz_j.txt 2>&1; tail -30 $O/fuzz_j.txt | cut -c1-300
