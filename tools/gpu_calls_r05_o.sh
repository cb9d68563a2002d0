#!/bin/bash
# (1) replay of fuzz seed 2000473 (a capsule robot at |qdd| = 266 with omega 4.8e-5: outside A-D at ETA = 2e-5; clause E now judged)
# (2) A/B of the culled pair loop: product = next trip's sphere record read a trip ahead + the range tests of a full chunk batched
#     (one LDS round trip per chunk); tools/diag/librmp2_triponly.so (-DRMP2_BATCHED_TESTS=0); tools/diag/librmp2_plainloop.so (neither)
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 120 python tools/fuzz_parity.py --seeds 2000473 2000474 --verbose > $O/fuzz_2000473.txt 2>&1; tail -3 $O/fuzz_2000473.txt | cut -c1-600
{ echo "# us per step: product (trip ahead + batched tests) | trip ahead only | neither"
for wl in config3 config3c config5 config4; do
  for lib in "" tools/diag/librmp2_triponly.so tools/diag/librmp2_plainloop.so; do
    if [ -n "$lib" ]; then export RMP2_LIB=$lib; else unset RMP2_LIB; fi
    python bench.py --workload $wl --steps 1000 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl'.ljust(10), ('$lib' or 'product').ljust(36), '%8.2f us' % (j['ms_per_step']*1e3), ' kernel %8.2f us' % (j['roofline'].get('kernel_ms', 0)*1e3))"
  done
done; } > $O/trip_ahead_ab.txt 2>&1
cat $O/trip_ahead_ab.txt | cut -c1-200
