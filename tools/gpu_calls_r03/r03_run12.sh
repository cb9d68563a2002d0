#!/bin/bash
# round 3, GPU call 12: native RCCL exchange (one C-ABI call per config-4 step)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt
for x in native torch; do timeout -k 10 300 python bench.py --workload config4 --exchange $x --no-cpu-baseline --no-secondary > $O/bench_config4_$x.json 2> $O/bench_config4_$x.err; python -c "
import json; d=json.load(open('$O/bench_config4_$x.json')); print('config4 $x', round(d['ms_per_step']*1e3,2), 'us; host issue', round(d['host_issue_ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2), d['result_check'])"; tail -2 $O/bench_config4_$x.err; done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload config4 --no-cpu-baseline --no-secondary > $O/bench_torchrun1.json 2>$O/torchrun.err; cut -c1-200 $O/bench_torchrun1.json; tail -3 $O/torchrun.err
timeout -k 10 600 python bench.py --workload config4 --emulate-world 8 > $O/emul_config4.json 2> $O/emul_config4.err && python -c "
import json; d=json.load(open('$O/emul_config4.json')); e=d['emulated_scaling']; print({k:v for k,v in e.items() if k!='per_rank'}); [print(r['rank'], r['us_per_step'], r['kernel_us']) for r in e['per_rank']]"
