#!/bin/bash
# round 3, GPU call 1: HEAD of round 2 on this round's box -- GPU tests, default bench line, config5 world-1 line
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err && cut -c1-600 $O/bench_default.json
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err && cut -c1-400 $O/bench_config5.json
