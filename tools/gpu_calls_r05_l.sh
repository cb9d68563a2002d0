#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_kernel_variants.py -q -m gpu -k "streamed" > $O/gpu_suite_l.log 2>&1; rc=$?; tail -5 $O/gpu_suite_l.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_calls_r05_f.sh
