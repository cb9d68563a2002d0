#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
for lib in riemannian_motion_policies_amd/librmp2_hip.so tools/diag/librmp2_prio.so; do
for w in 2 3 4; do
RMP2_LIB=$PWD/$lib RMP2_QUAD_MINW=$w timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$lib c3 65536 minw$w',j['ms_per_step'])"
done
RMP2_LIB=$PWD/$lib RMP2_QUAD_MINW=3 timeout -k 10 120 python bench.py --robots 262144 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$lib c3 262144 minw3',j['ms_per_step'])"
RMP2_LIB=$PWD/$lib RMP2_QUAD_MINW=3 timeout -k 10 120 python bench.py --robots 49152 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$lib c3 49152 minw3',j['ms_per_step'])"
done
