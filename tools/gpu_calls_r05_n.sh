#!/bin/bash
# interface B, two-wave form: the odd slot of every SIMD starts its first wave late (RMP2_STREAM_STAGGER=n, units of ~3.4 us)
O=gpurun_out/r05; mkdir -p $O
{ echo "# config3b, 65 536 robots, default two-wave form: us per step against the start offset of the odd wave slot (first round only)"
for s in 0 2 4 6 8 10 12 16; do
  RMP2_STREAM_STAGGER=$s python bench.py --workload config3b --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger $s'.ljust(14), '%8.2f us' % (j['ms_per_step']*1e3), ' rejected', j['result_check']['rejected'], ' [' + str(j['config'].get('kernel', ''))[:70] + ']')"
done
for s in 0 6; do
  RMP2_STREAM_STAGGER=$s python bench.py --workload config3b --robots 131072 --steps 50 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('131072 robots, stagger $s'.ljust(28), '%8.2f us' % (j['ms_per_step']*1e3))"
done; } > $O/interface_b_stagger_two_wave.txt 2>&1
cat $O/interface_b_stagger_two_wave.txt | cut -c1-200
