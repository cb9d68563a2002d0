#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -x -q -m gpu -k "explicit or pairs or golden or register_caps or lds_dma or full_size" > $O/win_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/win_tests.log; tail -4 $O/win_tests.log
[ $rc -eq 0 ] || exit $rc
for R in 65536 32768 131072 49152; do for w in 2 3 4; do
  RMP2_QUAD_MINW=$w python bench.py --workload config3b --robots $R --no-cpu-baseline --no-secondary --steps 800 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$R minw=$w:', round(j['ms_per_step']*1e3,2), 'us/step, hbm frac', round(j['roofline']['frac'],3), 'rejected', j['result_check']['rejected'])"
done; done
