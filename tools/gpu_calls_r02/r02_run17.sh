#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python bench.py --workload config4 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2>$O/bench_config4.err; cut -c1-330 $O/bench_config4.json
python tools/exchange_timing.py > $O/exchange_timing.txt 2>/dev/null; tail -8 $O/exchange_timing.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
