#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_random_robots.py -m gpu -x -q > $O/pytest_gpu_quad.txt 2>&1; tail -2 $O/pytest_gpu_quad.txt
for i in 1 2; do
timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary > $O/bench_c3.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3.json'));print('c3 65536',j['ms_per_step'],j['roofline']['kernel_ms'])"
done
timeout -k 10 120 python bench.py --robots 262144 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c3 262144',j['ms_per_step'])"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY --output-format csv -d $O/a -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_sq.py $O/a; rm -rf $O/a
