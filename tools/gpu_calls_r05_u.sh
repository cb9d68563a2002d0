#!/bin/bash
# link-geometry pair loop with the batched range tests: tests of the link modes, timing
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_capsules.py tests/test_gpu_exp05.py tests/test_cylinders.py tests/test_gpu_dropin.py -q -m gpu > $O/gpu_suite_u.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_u.log; tail -4 $O/gpu_suite_u.log | cut -c1-300
{ echo "# us per step, link-geometry loop: range tests batched ($NOTE)"
for wl in config3l config3 "config5 --link-geometry"; do
  python bench.py --workload $wl --steps 1000 --no-cpu-baseline --no-secondary 2>>$O/link_batched.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl'.ljust(26), '%8.2f us' % (j['ms_per_step']*1e3), ' kernel %8.2f us' % (j['roofline'].get('kernel_ms', 0)*1e3))"
done; } > $O/link_batched.txt 2>&1
cat $O/link_batched.txt
