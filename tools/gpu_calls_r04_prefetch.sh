#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_capsules.py tests/test_gpu_dropin.py -x -q -m gpu -k "explicit or pairs or golden or closest or link or lds_dma or experiment06 or full_size" > $O/pf_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pf_tests.log; tail -5 $O/pf_tests.log
[ $rc -eq 0 ] || exit $rc
for R in 65536 32768 131072; do for p in 1 0; do
  RMP2_EXPLICIT_PREFETCH=$p python bench.py --workload config3b --robots $R --no-cpu-baseline --no-secondary --steps 1000 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$R prefetch=$p:', round(j['ms_per_step']*1e3,2), 'us/step, hbm frac', round(j['roofline']['frac'],3), 'rejected', j['result_check']['rejected'])"
done; done
