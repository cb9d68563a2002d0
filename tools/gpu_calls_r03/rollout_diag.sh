#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dropin.py tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -x -q -m gpu > gpurun_out/quarantine_tests.log 2>&1 || { tail -40 gpurun_out/quarantine_tests.log; exit 1; }
tail -2 gpurun_out/quarantine_tests.log
python tools/rollout_diag.py 65536 > gpurun_out/rollout_diag.txt 2>&1 || { tail -30 gpurun_out/rollout_diag.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/rollout_diag.txt
