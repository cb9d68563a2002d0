#!/bin/bash
# round 3, GPU call 4: per-mode plain builds of the quad kernel + closed-form 2 x 2 resolve: GPU tests, flag tail, bench lines,
# emulated scaling again
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
timeout -k 10 600 python tools/flag_tail.py > $O/flag_tail.txt 2> $O/flag_tail.err; cat $O/flag_tail.txt; tail -3 $O/flag_tail.err
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
for w in 2 3 4; do RMP2_QUAD_MINW=$w timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 500 > $O/bench_minw$w.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_minw$w.json')); print('minw $w', d['ms_per_step']*1e3, 'us', d['roofline']['kernel_ms']*1e3)"; done
timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline > $O/bench_config3b.json 2> $O/bench_config3b.err; cut -c1-300 $O/bench_config3b.json; tail -2 $O/bench_config3b.err
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-300 $O/bench_config5.json
timeout -k 10 600 python bench.py --workload config5 --emulate-world 8 --calibrate --calibrate-robots 32768 --compare-flop-model > $O/emul_config5.json 2> $O/emul_config5.err && python -c "
import json; d=json.load(open('$O/emul_config5.json')); e=d['emulated_scaling']; print(json.dumps({k:v for k,v in e.items() if k not in ('per_rank',)}, indent=None)); [print(r) for r in e['per_rank']]"
timeout -k 10 600 python bench.py --workload config4 --emulate-world 8 > $O/emul_config4.json 2> $O/emul_config4.err && python -c "
import json; d=json.load(open('$O/emul_config4.json')); e=d['emulated_scaling']; print({k:v for k,v in e.items() if k!='per_rank'}); [print(r['rank'], r['us_per_step'], r['kernel_us']) for r in e['per_rank']]"
