#!/bin/bash
# round 3, GPU call 21: native exchange at depth 1 / 2
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k nccl > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
for d in 1 2; do for t in 500 0; do RMP2_EXCHANGE_THROTTLE_US=$t timeout -k 10 300 python bench.py --workload config4 --exchange native --exchange-depth $d --no-cpu-baseline --no-secondary > $O/b.json 2> $O/err; python -c "
import json; d=json.load(open('$O/b.json')); print('config4 native depth $d throttle $t:', round(d['ms_per_step']*1e3,2), 'us; host issue', round(d['host_issue_ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2), d['result_check']['within_tolerance'])"; done; done
timeout -k 10 600 python bench.py --workload config4 --emulate-world 8 > $O/emul4.json 2>$O/err; python -c "
import json; e=json.load(open('$O/emul4.json'))['emulated_scaling']; print('emulated (peer waits kept) depth 2:', [round(r['us_per_step'],1) for r in e['per_rank']])"
timeout -k 10 600 python bench.py --workload config4 --emulate-world 8 --exchange-depth 1 > $O/emul4.json 2>$O/err; python -c "
import json; e=json.load(open('$O/emul4.json'))['emulated_scaling']; print('emulated (peer waits kept) depth 1:', [round(r['us_per_step'],1) for r in e['per_rank']])"
