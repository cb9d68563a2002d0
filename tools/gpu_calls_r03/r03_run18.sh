#!/bin/bash
# round 3, GPU call 18: rollout with moving obstacle tables and with the strict pseudo-inverse
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
(python tools/rollout_timing.py config2 4096 50; RMP2_KERNEL=quad python tools/rollout_timing.py config2 4096 50; python tools/rollout_timing.py config3 4096 50; RMP2_KERNEL=quad python tools/rollout_timing.py config3 4096 50; RMP2_KERNEL=quad python tools/rollout_timing.py config3 65536 20) > $O/rollout.txt 2>/dev/null; cat $O/rollout.txt
