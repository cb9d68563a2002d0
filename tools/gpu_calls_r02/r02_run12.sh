#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
for R in 40960 57344 73728 81920 90112 114688 163840 196608 262144; do
for w in 2 3; do
RMP2_QUAD_MINW=$w timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('c3 $R b=%.1f minw$w' % ($R/16384.), j['ms_per_step'], j['value']/1e6)"
done; done
