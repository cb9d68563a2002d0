#!/bin/bash
# what happens between 1 024 and 1 025 waves (16 384 / 16 400 robots) in the quad mapping: config 3, solve = pinv, RMP2_KERNEL=quad
O=gpurun_out/r05; mkdir -p $O
{ echo "# us per step, config3 --solve pinv, RMP2_KERNEL=quad: rows = robots, columns = RMP2_QUAD_MINW unset / 2 / 4 (each with the latency build and with RMP2_QUAD_LATENCY_BLOCKS=0)"
for R in 8192 12288 16384 16400 20480 32768; do
  line="$R"
  for mw in "" 2 4; do for lb in "" 0; do
    if [ -n "$mw" ]; then export RMP2_QUAD_MINW=$mw; else unset RMP2_QUAD_MINW; fi
    if [ -n "$lb" ]; then export RMP2_QUAD_LATENCY_BLOCKS=$lb; else unset RMP2_QUAD_LATENCY_BLOCKS; fi
    v=$(RMP2_KERNEL=quad python bench.py --workload config3 --robots $R --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (j['ms_per_step']*1e3))")
    line="$line  minw=${mw:-auto}/lat=${lb:-def}:$v"
  done; done
  echo "$line"
done; } > $O/quad_1024_step.txt 2>&1
cat $O/quad_1024_step.txt
