#!/bin/bash
# the range tests in two batches of four (no spill in the step's prologue): timing of the workloads the batching moved
O=gpurun_out/r05; mkdir -p $O
{ echo "# us per step, range tests in two batches of four + row-joint records of a frame in one round trip + identity-leaf rows in one round trip (product)"
for wl in config3 config3c config5 config4 config2; do
  python bench.py --workload $wl --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl'.ljust(10), '%8.2f us' % (j['ms_per_step']*1e3), ' kernel %8.2f us' % (j['roofline'].get('kernel_ms', 0)*1e3))"
done; } > $O/batched_reads_2.txt 2>&1
cat $O/batched_reads_2.txt
