#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
for R in 49152 65536 98304 131072 262144; do
line="c3 R=$R"
for pt in 0 1 2 3; do
RMP2_PRIO_TAIL=$pt timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null
line="$line | tail$pt $(python -c "import json;j=json.load(open('$O/b.json'));print('%.2f' % (j['ms_per_step']*1e3))")"
done
echo "$line"
done
