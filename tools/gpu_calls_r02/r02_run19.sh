#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_random_robots.py tests/test_gpu_dropin.py -m gpu -x -q > $O/pytest_gpu_quad.txt 2>&1; tail -2 $O/pytest_gpu_quad.txt
for cfg in "65536" "65536" "49152" "262144"; do
timeout -k 10 120 python bench.py --robots $cfg --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c3 $cfg',j['ms_per_step'])"
done
timeout -k 10 120 python bench.py --workload config2 --robots 65536 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c2 65536',j['ms_per_step'], j['roofline']['kernel'])"
