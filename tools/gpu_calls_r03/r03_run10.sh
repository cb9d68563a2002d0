#!/bin/bash
# round 3, GPU call 10: identity leaves before the walk for the odd wave slots (A/B against a -DRMP2_IDENT_FIRST=0 build);
# ds_read instead of flat_load through the laundered offsets
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
for rep in 1 2; do
for lib in product noident; do
  if [ $lib = product ]; then unset RMP2_LIB; else export RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_$lib.so; fi
  for R in 65536 131072 262144; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 1000 --robots $R > $O/b.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b.json')); print('$lib R=$R', round(d['ms_per_step']*1e3,2), 'us; kernel', round(d['roofline']['kernel_ms']*1e3,2))"; done
done; done
unset RMP2_LIB
