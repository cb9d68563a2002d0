#!/bin/bash
# the quad mapping's LATENCY build (program staged in LDS, all registers; grids of at most one wave per SIMD) against the throughput
# builds (RMP2_QUAD_LATENCY_BLOCKS=0) at 8 208 .. 16 384 robots, round-5 kernels
O=gpurun_out/r05; mkdir -p $O
{ echo "# us per step: default (latency build up to 1 024 waves) | RMP2_QUAD_LATENCY_BLOCKS=0 (throughput builds)"
for wl in config3 config3c config2; do for R in 8208 12288 16384; do for lb in "" 0; do
  if [ -n "$lb" ]; then export RMP2_QUAD_LATENCY_BLOCKS=$lb; else unset RMP2_QUAD_LATENCY_BLOCKS; fi
  RMP2_KERNEL=quad python bench.py --workload $wl --robots $R --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl'.ljust(9), '$R'.rjust(6), ('latency blocks $lb' if '$lb' else 'default').ljust(18), '%8.2f us' % (j['ms_per_step']*1e3))"
done; done; done; } > $O/quad_latency_build_ab.txt 2>&1
cat $O/quad_latency_build_ab.txt
