#!/bin/bash
# round 3, GPU call 22: attached-point leaves in the quad mapping
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k exp05 > $O/pytest_exp05.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_exp05.txt
tail -15 $O/pytest_exp05.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
