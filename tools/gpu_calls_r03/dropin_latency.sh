#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dropin.py tests/test_gpu_exp05.py tests/test_gpu_capsules.py -x -q -m gpu > gpurun_out/dropin_tests.log 2>&1 || { tail -40 gpurun_out/dropin_tests.log; exit 1; }
tail -2 gpurun_out/dropin_tests.log
python tools/dropin_latency.py 7 300 > gpurun_out/dropin_latency.txt 2>&1 || { tail -30 gpurun_out/dropin_latency.txt; exit 1; }
python tools/dropin_latency.py 32 300 >> gpurun_out/dropin_latency.txt 2>&1
grep -v amdgpu.ids gpurun_out/dropin_latency.txt
