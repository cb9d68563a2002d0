#!/bin/bash
# 2-dof robots with link geometry under solve = pinv (lifted), the link loop back at batches of four: tests, a fuzz run, timings
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_capsules.py tests/test_gpu_exp05.py tests/test_cylinders.py tests/test_gpu_fuzz.py -q -m gpu > $O/gpu_suite_v.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_v.log; tail -4 $O/gpu_suite_v.log | cut -c1-300
timeout -k 10 300 python tools/fuzz_parity.py --seeds 3000000 3060000 --minutes 3.5 > $O/fuzz_v.json 2>&1; tail -40 $O/fuzz_v.json | grep -A3 '"passed"\|declined_reasons' | cut -c1-200
{ echo "# us per step"
for wl in "config3l" "config5 --link-geometry" "config5"; do
  python bench.py --workload $wl --steps 1000 --no-cpu-baseline --no-secondary 2>>$O/link_v.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl'.ljust(26), '%8.2f us' % (j['ms_per_step']*1e3), j['config'].get('solve'), {k: v.get('rejected') for k, v in j['result_check'].items() if isinstance(v, dict) and 'rejected' in v} or j['result_check'].get('rejected'))"
done; } > $O/link_v.txt 2>&1
cat $O/link_v.txt; tail -3 $O/link_v.err | cut -c1-300
