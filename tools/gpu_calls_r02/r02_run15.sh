#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
for lib in riemannian_motion_policies_amd/librmp2_hip.so tools/diag/librmp2_prio.so tools/diag/librmp2_prio2.so tools/diag/librmp2_prio3.so; do
for cfg in "65536 2" "65536 4" "49152 3" "131072 3" "262144 3" "32768 2"; do
set -- $cfg
RMP2_LIB=$PWD/$lib RMP2_QUAD_MINW=$2 timeout -k 10 120 python bench.py --robots $1 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$lib c3 $1 minw$2',j['ms_per_step'])"
done; done
