#!/bin/bash
# round 3, GPU call 24: interface B -- lane-local evaluation of the loaded pairs against the compacted + re-fetched loop (A/B)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "pairs or variants or parity or capsules" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
for lib in product explicit_noprefetch; do
  if [ $lib = product ]; then unset RMP2_LIB; else export RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_$lib.so; fi
  for w in 2; do for R in 32768 65536; do
  RMP2_QUAD_MINW=$w timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline --steps 500 --robots $R > $O/b.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b.json')); r=d['roofline']; print('$lib minw $w R=$R', round(d['ms_per_step']*1e3,1), 'us; hbm frac', round(r['frac'],3), d['result_check']['within_tolerance'])"; done; done
done
unset RMP2_LIB
