#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_random_robots.py tests/test_gpu_dropin.py tests/test_gpu_capsules.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for sym in 0 1; do
for cfg in "65536 0" "65536 0" "49152 0" "262144 0" "65536 4" "16384 0"; do
set -- $cfg
RMP2_QUAD_SYM=$sym RMP2_QUAD_MINW=$2 timeout -k 10 120 python bench.py --robots $1 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('sym=$sym c3 $1 minw$2',round(j['ms_per_step']*1e3,2))"
done; done
