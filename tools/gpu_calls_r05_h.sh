#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/gpu_suite_h.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite_h.log; tail -12 $O/gpu_suite_h.log | cut -c1-300
timeout -k 10 400 python tools/fuzz_parity.py --seeds 500000 503000 --minutes 4 --log $O/fuzz_h.log > $O/fuzz_h.txt 2>&1; tail -40 $O/fuzz_h.txt | cut -c1-400
