#!/bin/bash
# the quad mapping's latency build (<= 1 024 waves) against the throughput build on the same grids: tuning build of the same sources
# (tools/diag/librmp2_tuning.so, -DRMP2_TUNING: RMP2_QUAD_LATENCY_BLOCKS is read), RMP2_KERNEL=quad, solve = pinv
O=gpurun_out/r05; mkdir -p $O
export RMP2_LIB=tools/diag/librmp2_tuning.so
{ echo "# us per step: rows = workload, robots; latency build (default) | throughput build (RMP2_QUAD_LATENCY_BLOCKS=0)"
for wl in config3 config2 config3c; do for R in 1024 2048 4096 8192 12288 16384; do
  line="$wl $R"
  for lb in "" 0; do
    if [ -n "$lb" ]; then export RMP2_QUAD_LATENCY_BLOCKS=$lb; else unset RMP2_QUAD_LATENCY_BLOCKS; fi
    v=$(RMP2_KERNEL=quad python bench.py --workload $wl --robots $R --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (j['ms_per_step']*1e3))")
    line="$line  ${lb:-latency}:$v"
  done
  echo "$line"
done; done; } > $O/quad_latency_build_ab2.txt 2>&1
cat $O/quad_latency_build_ab2.txt
