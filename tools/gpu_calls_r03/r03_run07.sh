#!/bin/bash
# round 3, GPU call 7: culled explicit-pair loop (interface B)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
for w in 2 3 4; do RMP2_QUAD_MINW=$w timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline --steps 500 > $O/bench_config3b_minw$w.json 2> $O/bench_config3b.err; python -c "
import json; d=json.load(open('$O/bench_config3b_minw$w.json')); r=d['roofline']; print('config3b minw $w', d['ms_per_step']*1e3, 'us; kernel', r['kernel_ms']*1e3, 'hbm frac', r['frac'], 'GB/s', r['achieved'], d['result_check'])"; done
for R in 16384 32768 131072; do timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline --steps 300 --robots $R > $O/bench_config3b_R$R.json 2> $O/bench_config3b.err; python -c "
import json; d=json.load(open('$O/bench_config3b_R$R.json')); r=d['roofline']; print('config3b R $R', d['ms_per_step']*1e3, 'us; kernel', r['kernel_ms']*1e3, 'hbm frac', r['frac'], 'GB/s', r['achieved'])"; done
