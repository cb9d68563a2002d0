#!/bin/bash
mkdir -p gpurun_out
for R in 64 4096 65536; do
  for k in auto lane; do
    if [ $k = lane ]; then export RMP2_KERNEL=lane; else unset RMP2_KERNEL; fi
    python bench.py --solve pinv --robots $R --no-cpu-baseline --no-secondary --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('solve=pinv R=$R RMP2_KERNEL=$k', round(d['ms_per_step']*1e3,1), 'us per step;', d['roofline']['kernel'][:70])"
  done
done
