#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
timeout -k 10 120 python bench.py --no-cpu-baseline > $O/bench_c3.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3.json'));print('c3 65536',j['ms_per_step'],'| c2 4096',j['secondary']['ms_per_step'],j['secondary']['roofline']['kernel_ms'])"
timeout -k 10 120 python bench.py --robots 4096 --no-cpu-baseline --no-secondary > $O/bench_c3_4k.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3_4k.json'));print('c3 4096',j['ms_per_step'])"
