#!/bin/bash
# interface B (explicit pairs): LDS-DMA stream against register loads, parity first
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_capsules.py -x -q -m gpu -k "explicit or pairs or golden or closest or link or lds_dma" > $O/glds_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/glds_tests.log; tail -5 $O/glds_tests.log
[ $rc -eq 0 ] || exit $rc
for R in 65536 32768 131072; do
  for g in 1 0; do  # (1: the LDS-DMA stream, opt-in)
    RMP2_EXPLICIT_GLDS=$g python bench.py --workload config3b --robots $R --no-cpu-baseline --no-secondary --steps 1000 > $O/b3b_g${g}_R$R.json 2> $O/b3b_g${g}_R$R.err || exit 1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/b3b_g*_R*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(j['ms_per_step']*1e3,2),'us/step | kernel', round(j['roofline']['kernel_ms']*1e3,2), '| hbm frac', round(j['roofline']['frac'],3), '| rejected', j['result_check']['rejected'])
PY
