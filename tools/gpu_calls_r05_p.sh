#!/bin/bash
# (1) fuzz seed 2000473, robot 106 against the fp32 envelope and leaf by leaf; (2) product (batched range tests, no trip-ahead) against the
# two diagnostic builds of the previous call
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 200 python tools/diag_envelope.py 2000473 106 > $O/diag_2000473.txt 2>&1; cat $O/diag_2000473.txt | cut -c1-300 | tail -14
timeout -k 10 300 python tools/diag_fuzz_leaf.py 2000473 106 > $O/diag_leaf_2000473.txt 2>&1; tail -12 $O/diag_leaf_2000473.txt | cut -c1-300
{ echo "# us per step: product (batched range tests) | trip ahead only | neither"
for wl in config3 config3c config5 config2; do
  for lib in "" tools/diag/librmp2_triponly.so tools/diag/librmp2_plainloop.so; do
    if [ -n "$lib" ]; then export RMP2_LIB=$lib; else unset RMP2_LIB; fi
    python bench.py --workload $wl --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl'.ljust(10), ('$lib' or 'product').ljust(36), '%8.2f us' % (j['ms_per_step']*1e3), ' kernel %8.2f us' % (j['roofline'].get('kernel_ms', 0)*1e3))"
  done
done; } > $O/trip_ahead_ab2.txt 2>&1
cat $O/trip_ahead_ab2.txt | cut -c1-200
