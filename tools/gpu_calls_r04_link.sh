#!/bin/bash
# link geometry fused into the step: lean builds (segments formed in the walk) against the attached-record builds
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_capsules.py tests/test_gpu_dropin.py -x -q -m gpu > $O/link_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/link_tests.log; tail -6 $O/link_tests.log
[ $rc -eq 0 ] || exit $rc
{
  echo "# config3l (65 536 Pandas, 8 link capsules fitted to the meshes x 32 spheres, pairs formed inside the step): us per step"
  for lean in 1 0; do for w in 3 2; do
    [ $lean = 0 ] && [ $w = 3 ] && continue
    RMP2_LINK_LEAN=$lean RMP2_QUAD_MINW=$w python bench.py --workload config3l --no-cpu-baseline --no-secondary --steps 1000 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean=$lean minw=$w:', round(j['ms_per_step']*1e3,2), 'us/step |', j['roofline']['kernel'][:90], '| rejected', j['result_check']['rejected'])"
  done; done
  python bench.py --workload config3l --no-cpu-baseline --no-secondary --steps 1000 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default:', round(j['ms_per_step']*1e3,2), 'us/step |', j['roofline']['kernel'][:90])"
} > $O/link_geometry.txt
cat $O/link_geometry.txt
