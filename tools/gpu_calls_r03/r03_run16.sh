#!/bin/bash
# round 3, GPU call 16: the pass loop as two instantiations (no SGPR spills, no hot VGPR spills): tests, bench, register-cap sweep
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('config3', round(d['ms_per_step']*1e3,2), 'us', d['value']/1e6, 'M steps/s; valu frac', d['roofline']['frac'], 'secondary', round(d['secondary']['ms_per_step']*1e3,2))"
timeout -k 10 600 python tools/minw_sweep.py config3 > $O/minw_config3.txt 2>&1; cat $O/minw_config3.txt
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_config5.json')); print('config5', round(d['ms_per_step']*1e3,2), 'us; kernel', round(d['roofline']['kernel_ms']*1e3,2))"
timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline > $O/bench_config3b.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_config3b.json')); print('config3b', round(d['ms_per_step']*1e3,2), 'us; hbm frac', d['roofline']['frac'])"
timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_config4.json')); print('config4', round(d['ms_per_step']*1e3,2), 'us; kernel', round(d['roofline']['kernel_ms']*1e3,2))"
