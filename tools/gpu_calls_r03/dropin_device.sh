#!/bin/bash
# the device-resident experiment-06 loop + the whole drop-in file
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dropin.py tests/test_gpu_exp05.py -x -q -m gpu > gpurun_out/dropin_device.log 2>&1 || { tail -40 gpurun_out/dropin_device.log; exit 1; }
tail -5 gpurun_out/dropin_device.log
