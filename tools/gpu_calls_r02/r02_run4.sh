#!/bin/bash
# round 2, GPU call: culled pair loop -- parity (all mappings), bench, counters
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_dropin.py tests/test_gpu_capsules.py -m gpu -x -q > $O/pytest_gpu_quad.txt 2>&1; tail -5 $O/pytest_gpu_quad.txt
timeout -k 10 120 python bench.py --no-cpu-baseline > $O/bench_c3.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3.json'));print('c3 65536',j['ms_per_step'],j['roofline']['kernel_ms'],j['roofline']['valu']['frac'],'| c2',j['secondary']['ms_per_step'])"
RMP2_QUAD_MINW=4 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary > $O/bench_c3_minw4.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3_minw4.json'));print('c3 65536 minw4',j['ms_per_step'])"
timeout -k 10 120 python bench.py --robots 4096 --no-cpu-baseline --no-secondary > $O/bench_c3_4k.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3_4k.json'));print('c3 4096',j['ms_per_step'],j['roofline']['kernel'])"
timeout -k 10 120 python bench.py --workload config5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c5.json'));print('c5',j['ms_per_step'],j['roofline']['kernel_ms'])"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/a -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_sq.py $O/a; rm -rf $O/a
