#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
for w in 2 3 4; do
RMP2_KERNEL=quad RMP2_QUAD_MINW=$w timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_dropin.py tests/test_gpu_capsules.py tests/test_gpu_random_robots.py -m gpu -x -q > $O/pytest_gpu_quad$w.txt 2>&1; tail -2 $O/pytest_gpu_quad$w.txt
done
for R in 32768 49152 65536 131072 262144; do
timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b_$R.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b_$R.json'));print('c3 $R auto',j['ms_per_step'], j['value']/1e6)"
done
timeout -k 10 120 python bench.py --workload config2 --robots 262144 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c2 262144 auto',j['ms_per_step'], j['value']/1e6, j['roofline']['kernel'])"
