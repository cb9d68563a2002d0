#!/bin/bash
# A/B of the native exchange: tables one / two steps ahead, two or three table buffers in rotation (RMP2_EXCHANGE_BUFFERS=3),
# plain world 1 and with the GPU-side wait an N-rank run keeps (--emulate-world 2: peer_wait)
O=gpurun_out/r04; mkdir -p $O
for depth in 1 2; do for bufs in 2 3; do
  RMP2_EXCHANGE_BUFFERS=$bufs python bench.py --workload config4 --exchange-depth $depth --no-cpu-baseline --no-secondary --steps 3000 > $O/x_d${depth}_b${bufs}.json 2>$O/x_d${depth}_b${bufs}.err || exit 1
  RMP2_EXCHANGE_BUFFERS=$bufs python bench.py --workload config4 --exchange-depth $depth --emulate-world 2 --steps 500 > $O/xe_d${depth}_b${bufs}.json 2>$O/xe_d${depth}_b${bufs}.err || exit 1
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/x*_d*_b*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    if 'emulated_scaling' in j: print(f, [round(r['us_per_step'],2) for r in j['emulated_scaling']['per_rank']])
    else: print(f, round(j['ms_per_step']*1e3,2), 'us/step; kernel', round(j['roofline']['kernel_ms']*1e3,2))
PY
