#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite_chk.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_chk.log; tail -4 $O/gpu_suite_chk.log | cut -c1-300
{ echo "# us per step, solve = pinv: hex | quad | default dispatch"
for wl in config3 config2; do for R in 4096 8192 8208 12288 16384; do
  line="$wl $R"
  for k in hex quad ""; do
    if [ -n "$k" ]; then export RMP2_KERNEL=$k; else unset RMP2_KERNEL; fi
    v=$(python bench.py --workload $wl --robots $R --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (j['ms_per_step']*1e3))")
    line="$line  ${k:-auto}:$v"
  done
  echo "$line"
done; done
unset RMP2_KERNEL
python bench.py --workload config5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 %.2f us' % (j['ms_per_step']*1e3))"
} > $O/dispatch_after_latency.txt 2>&1
cat $O/dispatch_after_latency.txt
