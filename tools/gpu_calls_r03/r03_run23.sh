#!/bin/bash
# round 3, GPU call 23: two robot types in one grid (rmp2_step_pair)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2>$O/err; python -c "
import json; d=json.load(open('$O/bench_config5.json')); print('config5', round(d['ms_per_step']*1e3,2), 'us; host', round(d['host_issue_ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2), d['roofline']['kernel'][:60])"; tail -2 $O/err
timeout -k 10 300 python bench.py --workload config5 --robots 65536 --no-cpu-baseline > $O/bench_config5_64k.json 2>$O/err; python -c "
import json; d=json.load(open('$O/bench_config5_64k.json')); print('config5 65536', round(d['ms_per_step']*1e3,2), 'us', round(d['value']/1e6), 'M/s')"
