#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k nccl > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
for t in 500 0; do RMP2_EXCHANGE_THROTTLE_US=$t timeout -k 10 300 python bench.py --workload config4 --exchange native --no-cpu-baseline --no-secondary > $O/bench_config4_native_t$t.json 2> $O/err; python -c "
import json; d=json.load(open('$O/bench_config4_native_t$t.json')); print('config4 native throttle $t', round(d['ms_per_step']*1e3,2), 'us; host issue', round(d['host_issue_ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2), d['result_check'])"; done
timeout -k 10 600 python bench.py --workload config4 --emulate-world 8 > $O/emul4.json 2>$O/err; python -c "import json; e=json.load(open('$O/emul4.json'))['emulated_scaling']; print([round(r['us_per_step'],1) for r in e['per_rank']])"; timeout -k 10 300 python bench.py --workload config4 --exchange torch --no-cpu-baseline --no-secondary > $O/bench_config4_torch.json 2> $O/err; python -c "
import json; d=json.load(open('$O/bench_config4_torch.json')); print('config4 torch', round(d['ms_per_step']*1e3,2), 'us; host issue', round(d['host_issue_ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2))"
