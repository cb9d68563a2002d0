#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_exp05.py tests/test_gpu_dropin.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests_b.log 2>&1; echo "pytest rc=$?" >> $O/tests_b.log; tail -30 $O/tests_b.log
python tools/diag_strict.py 65536 2>/dev/null | tail -6
