#!/bin/bash
# interface B, streamed form: start offset between the four waves of a SIMD (RMP2_STREAM_STAGGER, units of 3.4 us)
O=gpurun_out/r05; mkdir -p $O
{
  echo "# config3b, 65 536 robots, us per step; s0 = single-loop two-wave form; stagger n = streamed form, waves of a SIMD n x 3.4 us apart"
  RMP2_EXPLICIT_STREAM=0 python bench.py --workload config3b --no-cpu-baseline --no-secondary --steps 600 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single-loop:', round(j['ms_per_step']*1e3,2), 'us/step, hbm frac', round(j['roofline']['frac'],3))"
  for g in 0 1 2 3 4 6; do
    RMP2_EXPLICIT_STREAM=1 RMP2_STREAM_STAGGER=$g python bench.py --workload config3b --no-cpu-baseline --no-secondary --steps 600 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger $g:', round(j['ms_per_step']*1e3,2), 'us/step, hbm frac', round(j['roofline']['frac'],3), j['result_check']['admitted_by']['A_north_star_1e-5'])"
  done
} > $O/interface_b_stagger.txt 2>&1
cat $O/interface_b_stagger.txt | cut -c1-200
