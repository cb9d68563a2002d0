#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_fuzz.py tests/test_gpu_accuracy_envelope.py -q > $O/gpu_suite_s.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_s.log; tail -5 $O/gpu_suite_s.log | cut -c1-300
