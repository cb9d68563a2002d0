#!/bin/bash
# round 3, GPU call 20: time curves per robot type, curve-based cut of the mixed fleet, emulated 8-rank scaling, shard-shape tests
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 600 python tools/calibrate_costs.py > $O/cost_calibration.json 2> $O/err; cat $O/cost_calibration.json | tr -d '\n' | cut -c1-1500; echo
timeout -k 10 600 python bench.py --workload config5 --emulate-world 8 --calibrate --compare-flop-model > $O/emul_config5.json 2> $O/emul_config5.err && python -c "
import json; d=json.load(open('$O/emul_config5.json')); e=d['emulated_scaling']; print({k:v for k,v in e.items() if k not in ('per_rank','cost_model','plans')}); print(e.get('plans')); [print(r) for r in e['per_rank']]"
tail -3 $O/emul_config5.err
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "shard or fleet" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
