#!/bin/bash
# round 3, GPU call 19: closest-point stage with link geometry
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt
