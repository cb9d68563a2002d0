#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
for lib in riemannian_motion_policies_amd/librmp2_hip.so tools/diag/librmp2_prioA.so riemannian_motion_policies_amd/librmp2_hip.so tools/diag/librmp2_prioA.so; do
for R in 49152 65536 262144; do
RMP2_LIB=$PWD/$lib timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$lib c3 $R',round(j['ms_per_step']*1e3,2))"
done; done
