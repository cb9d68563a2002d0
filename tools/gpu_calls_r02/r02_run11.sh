#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
RMP2_KERNEL=quad RMP2_QUAD_MINW=3 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -m gpu -x -q > $O/pytest_gpu_quad3.txt 2>&1; tail -3 $O/pytest_gpu_quad3.txt
for R in 32768 49152 65536 98304 131072; do
for w in 2 3 4; do
RMP2_QUAD_MINW=$w timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c3 $R minw$w',j['ms_per_step'], j['value']/1e6)"
done; done
