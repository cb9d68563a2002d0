#!/bin/bash
# round 3, GPU call 9: interface B with cross-leaf prefetch
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "pairs or variants or parity" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
for R in 32768 65536 131072; do timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline --steps 500 --robots $R > $O/bench_config3b_R$R.json 2> $O/bench_config3b.err; python -c "
import json; d=json.load(open('$O/bench_config3b_R$R.json')); r=d['roofline']; print('config3b R $R', d['ms_per_step']*1e3, 'us; kernel', r['kernel_ms']*1e3, 'hbm frac', r['frac'], 'GB/s', r['achieved'], d['result_check'])"; done
