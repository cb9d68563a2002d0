#!/bin/bash
# sets without distance leaves (config 2) at fleet size: quad against lane-per-robot against the default dispatch, solve = pinv and auto
O=gpurun_out/r05; mkdir -p $O
{ echo "# us per step, config 2: hex | quad | lane | default dispatch"
for sv in pinv auto; do for R in 16384 32768 32784 65536 131072; do
  line="$sv $R"
  for k in quad lane ""; do
    if [ -n "$k" ]; then export RMP2_KERNEL=$k; else unset RMP2_KERNEL; fi
    v=$(python bench.py --workload config2 --solve $sv --robots $R --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (j['ms_per_step']*1e3))")
    line="$line  ${k:-auto}:$v"
  done
  echo "$line"
done; done; } > $O/dispatch_lane_quad.txt 2>&1
cat $O/dispatch_lane_quad.txt
