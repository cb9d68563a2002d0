#!/bin/bash
mkdir -p gpurun_out
for lib in main pipe0; do
  if [ $lib = main ]; then unset RMP2_LIB; else export RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_pipe0.so; fi
  for R in 65536 32768; do
    python bench.py --workload config3b --robots $R --no-cpu-baseline --no-secondary --steps 500 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib R=$R', round(d['ms_per_step']*1e3,1), 'us', d['result_check'])"
  done
done
