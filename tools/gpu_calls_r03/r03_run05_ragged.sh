#!/bin/bash
# round 3, GPU call 5: ragged lists as membership masks; register-cap and mapping sweeps with the spill-free builds
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-300 $O/bench_config5.json
timeout -k 10 600 python tools/minw_sweep.py config3 > $O/minw_config3.txt 2>&1; cat $O/minw_config3.txt
timeout -k 10 600 python tools/minw_sweep.py config3r 16384 21760 24576 32768 49152 65536 > $O/minw_config3r.txt 2>&1; cat $O/minw_config3r.txt
timeout -k 10 600 python tools/dispatch_sweep.py tj5 > $O/dispatch_tj5.txt 2>&1; cat $O/dispatch_tj5.txt
timeout -k 10 600 python tools/dispatch_sweep.py config3r 4096 8192 12288 16384 20480 > $O/dispatch_config3r.txt 2>&1; cat $O/dispatch_config3r.txt
