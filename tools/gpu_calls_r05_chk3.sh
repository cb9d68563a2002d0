#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite_chk3.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_chk3.log; tail -4 $O/gpu_suite_chk3.log | cut -c1-300
for a in "--solve pinv --robots 65536" "--solve auto --robots 40000" "--solve auto --robots 65536"; do python bench.py --workload config2 $a --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 $a', '%.2f us' % (j['ms_per_step']*1e3))"; done
