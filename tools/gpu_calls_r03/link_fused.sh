#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_capsules.py -x -q -m gpu > gpurun_out/link_fused_tests.log 2>&1 || { tail -60 gpurun_out/link_fused_tests.log; exit 1; }
tail -3 gpurun_out/link_fused_tests.log
python tools/closest_stage_timing.py 65536 50 > gpurun_out/closest_stage.txt 2>&1 || { tail -30 gpurun_out/closest_stage.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/closest_stage.txt
