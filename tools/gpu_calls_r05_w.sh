#!/bin/bash
# hex resolve: pivot rows and the certificate's back substitution by DPP row broadcasts instead of LDS exchanges
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/gpu_suite_w.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_w.log; tail -4 $O/gpu_suite_w.log | cut -c1-300
{ echo "# us per step, hex mapping, pivot rows by DPP ($NOTE)"
for a in "config2 --solve pinv" "config2 --solve auto" "config2 --solve pinv --robots 1024" "config2 --solve pinv --robots 8192" "config3 --solve pinv --robots 4096"; do
  python bench.py --workload $a --steps 2000 --no-cpu-baseline --no-secondary 2>>$O/hex_w.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a'.ljust(40), '%8.2f us' % (j['ms_per_step']*1e3), ' kernel %8.2f us' % (j['roofline'].get('kernel_ms', 0)*1e3), j['config'].get('kernel', '')[:40])"
done; } > $O/hex_dpp_$TAG.txt 2>&1
cat $O/hex_dpp_$TAG.txt
