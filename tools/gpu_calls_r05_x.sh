#!/bin/bash
# the same timings with the LDS-exchange resolve (tools/diag/librmp2_hexlds.so: -DRMP2_HEX_DPP_PIVOTS=0), then the product again
O=gpurun_out/r05; mkdir -p $O
for lib in tools/diag/librmp2_hexlds.so ""; do
  if [ -n "$lib" ]; then export RMP2_LIB=$lib; TAG=lds; else unset RMP2_LIB; TAG=dpp; fi
  { echo "# us per step, hex mapping ($TAG)"
  for a in "config2 --solve pinv" "config2 --solve auto" "config2 --solve pinv --robots 1024" "config2 --solve pinv --robots 8192" "config3 --solve pinv --robots 4096"; do
    python bench.py --workload $a --steps 2000 --no-cpu-baseline --no-secondary 2>>$O/hex_x.err | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a'.ljust(40), '%8.2f us' % (j['ms_per_step']*1e3), ' kernel %8.2f us' % (j['roofline'].get('kernel_ms', 0)*1e3))"
  done; } > $O/hex_resolve_$TAG.txt 2>&1
  cat $O/hex_resolve_$TAG.txt
done
