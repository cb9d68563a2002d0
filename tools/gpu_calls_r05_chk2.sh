#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dropin.py -q > $O/gpu_suite_chk2.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_chk2.log; tail -4 $O/gpu_suite_chk2.log | cut -c1-300
