#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for R in 65536 32768; do for w in 2 3 4; do for g in 1 0; do
  RMP2_QUAD_MINW=$w RMP2_EXPLICIT_GLDS=$g python bench.py --workload config3b --robots $R --no-cpu-baseline --no-secondary --steps 600 > $O/b3b_w${w}_g${g}_R$R.json 2> $O/b3b_w${w}_g${g}_R$R.err || exit 1
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/b3b_w*_g*_R*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(j['ms_per_step']*1e3,2),'us/step | hbm frac', round(j['roofline']['frac'],3), '| rejected', j['result_check']['rejected'])
PY
