#!/bin/bash
# the ragged-list route of link geometry beyond the fused limits: its tests, then the fuzz seeds the campaign used to decline
# (3000669, 3001226, 3001231, 3001263 in profiles/r05_fuzz_parity.json's run)
O=gpurun_out/r05/ragged; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_capsules.py tests/test_cylinders.py -q -m gpu -x -s > $O/tests.log 2>&1; rc=$?; grep "fused vs staged" $O/tests.log; tail -5 $O/tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
for r in "3000660 3000680" "3001220 3001270"; do
  n=$(echo $r | cut -d' ' -f1)
  timeout -k 10 120 python tools/fuzz_parity.py --seeds $r --minutes 1 --log $O/fuzz_$n.log > $O/fuzz_$n.json 2>&1
  grep -E "\"cases\"|\"passed\"|\"declined\"|\"failed\"" $O/fuzz_$n.json | tr -d '\n'; echo
done
