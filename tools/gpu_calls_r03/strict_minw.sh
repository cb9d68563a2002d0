#!/bin/bash
mkdir -p gpurun_out
for w in 2 3 4; do
  RMP2_QUAD_MINW=$w python bench.py --solve pinv --no-cpu-baseline --no-secondary --steps 500 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('minw $w', round(d['ms_per_step']*1e3,1), 'us')"
done
