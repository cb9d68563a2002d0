#!/bin/bash
# round 3, GPU call 17: capsule tables on the culled / symmetric / plain path
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
for w in config3 config3c; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-secondary > $O/bench_$w.json 2> $O/err_$w; python -c "
import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step']*1e3,2), 'us', d['roofline']['kernel'][:60], d['result_check'])"; tail -2 $O/err_$w; done
