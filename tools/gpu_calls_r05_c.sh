#!/bin/bash
# round 5, third call: streamed explicit pairs (interface B) -- parity, then timing against the single-loop form; FK error diagnostic
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_accuracy_envelope.py tests/test_gpu_dropin.py -q -m gpu -x > $O/gpu_suite_c.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite_c.log; tail -12 $O/gpu_suite_c.log
python tools/diag_fk_error.py 2>/dev/null | cut -c1-400 > $O/diag_fk_error.txt; cat $O/diag_fk_error.txt
[ $rc -eq 0 ] || exit $rc
{
  echo "# interface B (config3b): us per step / fraction of 8 TB/s; s0 = RMP2_EXPLICIT_STREAM=0 (single-loop two-wave form), s1 = streamed"
  for R in 32768 49152 65536 81920 131072; do for g in 0 1; do
    RMP2_EXPLICIT_STREAM=$g timeout -k 10 200 python bench.py --workload config3b --robots $R --no-cpu-baseline --no-secondary --steps 1000 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$R s$g:', round(j['ms_per_step']*1e3,2), 'us/step, hbm frac', round(j['roofline']['frac'],3), j['roofline']['kernel'][:90], j['result_check']['admitted_by'])"
  done; done
} > $O/interface_b_stream.txt
cat $O/interface_b_stream.txt
