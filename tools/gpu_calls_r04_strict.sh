#!/bin/bash
# strict (solve = pinv) step: certifying one-launch form against the two-kernel all-Jacobi form
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_exp05.py tests/test_gpu_dropin.py tests/test_gpu_random_robots.py -x -q -m gpu -k "strict or pinv or exp05 or rollout or random" -s > $O/strict_tests.log 2>&1; echo "pytest rc=$?" >> $O/strict_tests.log; tail -25 $O/strict_tests.log
for R in 65536 64; do
  python bench.py --workload config3 --solve pinv --robots $R --no-cpu-baseline --no-secondary --steps 2000 > $O/strict_cert_R$R.json 2> $O/strict_cert_R$R.err || exit 1
  RMP2_STRICT_CERTIFY=0 python bench.py --workload config3 --solve pinv --robots $R --no-cpu-baseline --no-secondary --steps 1000 > $O/strict_jacobi_R$R.json 2> $O/strict_jacobi_R$R.err || exit 1
  python bench.py --workload config3 --robots $R --no-cpu-baseline --no-secondary --steps 2000 > $O/auto_R$R.json 2> $O/auto_R$R.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/strict_*_R*.json')+glob.glob('gpurun_out/r04/auto_R*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(j['ms_per_step']*1e3,2), 'us/step |', j['roofline']['kernel'][:100], '|', j['result_check']['admitted_by'], j['result_check']['rejected'])
PY
