#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cylinders.py tests/test_gpu_fuzz.py -q -m gpu > $O/gpu_suite_i.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite_i.log; tail -8 $O/gpu_suite_i.log | cut -c1-300
timeout -k 10 400 python tools/fuzz_parity.py --seeds 500000 503000 --minutes 3 --log $O/fuzz_i.log > $O/fuzz_i.txt 2>&1; tail -32 $O/fuzz_i.txt | cut -c1-300
