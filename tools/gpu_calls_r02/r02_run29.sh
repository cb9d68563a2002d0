#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_random_robots.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -2
for R in 16384 32768 49152 65536 65536 131072 262144; do
timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c3 $R auto',round(j['ms_per_step']*1e3,2), round(j['value']/1e6,1))"
done
