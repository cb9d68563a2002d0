#!/bin/bash
# round 3, GPU call 2: split library through the GPU tests; cost calibration; emulated 8-rank scaling of configs 5 / 4;
# first config3b (interface B) line; default bench line with the new roofline object
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
timeout -k 10 300 python tools/calibrate_costs.py 8192 16384 32768 65536 > $O/cost_calibration.json 2> $O/cost_calibration.err && cat $O/cost_calibration.json
timeout -k 10 600 python bench.py --workload config5 --emulate-world 8 --calibrate --calibrate-robots 32768 --compare-flop-model > $O/emul_config5.json 2> $O/emul_config5.err && python -c "
import json; d=json.load(open('$O/emul_config5.json')); e=d['emulated_scaling']; print(json.dumps({k:v for k,v in e.items() if k!='per_rank'}, indent=1)); [print(r) for r in e['per_rank']]"
timeout -k 10 600 python bench.py --workload config4 --emulate-world 8 > $O/emul_config4.json 2> $O/emul_config4.err && python -c "
import json; d=json.load(open('$O/emul_config4.json')); e=d['emulated_scaling']; print({k:v for k,v in e.items() if k!='per_rank'}); [print(r) for r in e['per_rank']]"
timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline > $O/bench_config3b.json 2> $O/bench_config3b.err && cut -c1-1500 $O/bench_config3b.json
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err && cut -c1-2500 $O/bench_default.json
