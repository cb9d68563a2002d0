#!/bin/bash
# round 2, GPU call: walk state zero-initialised -- quick parity, MINW=2 vs MINW=4 A/B
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -m gpu -x -q > $O/pytest_gpu_quad.txt 2>&1; tail -3 $O/pytest_gpu_quad.txt
RMP2_KERNEL=quad RMP2_QUAD_MINW=4 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -m gpu -x -q > $O/pytest_gpu_quad4.txt 2>&1; tail -3 $O/pytest_gpu_quad4.txt
for i in 1 2; do
timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary > $O/bench_c3.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3.json'));print('c3 65536 minw2',j['ms_per_step'],j['roofline']['kernel_ms'])"
RMP2_QUAD_MINW=4 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary > $O/bench_c3_minw4.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3_minw4.json'));print('c3 65536 minw4',j['ms_per_step'],j['roofline']['kernel_ms'])"
done
for R in 32768 131072 262144; do
timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c3 $R minw2',j['ms_per_step'])"
RMP2_QUAD_MINW=4 timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c3 $R minw4',j['ms_per_step'])"
done
RMP2_QUAD_MINW=4 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/a -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_sq.py $O/a; rm -rf $O/a
