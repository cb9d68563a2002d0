#!/bin/bash
# round 3, GPU call 6: tests; config-5 step as a HIP graph; new cuts
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
for g in "" "--no-graph"; do timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline $g > $O/bench_config5$g.json 2> $O/bench_config5$g.err; python -c "
import json; d=json.load(open('$O/bench_config5$g.json')); print('config5 $g', d['ms_per_step']*1e3, 'us; host issue', d['host_issue_ms_per_step']*1e3, d.get('step_issue'), d['roofline']['kernel_ms']*1e3)"; done
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
timeout -k 10 600 python bench.py --workload config5 --emulate-world 8 --calibrate --calibrate-robots 32768 > $O/emul_config5.json 2> $O/emul_config5.err && python -c "
import json; d=json.load(open('$O/emul_config5.json')); e=d['emulated_scaling']; print(json.dumps({k:v for k,v in e.items() if k not in ('per_rank',)}, indent=None)); [print(r) for r in e['per_rank']]"
