#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
for lib in riemannian_motion_policies_amd/librmp2_hip.so tools/diag/librmp2_maxilp.so tools/diag/librmp2_iterilp.so tools/diag/librmp2_minreg.so; do
for cfg in "config3 65536 2" "config3 65536 4" "config3 49152 3" "config2 4096 2" "config3 4096 2"; do
set -- $cfg
RMP2_LIB=$PWD/$lib RMP2_QUAD_MINW=$3 timeout -k 10 120 python bench.py --workload $1 --robots $2 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$lib $1 $2 minw$3',round(j['ms_per_step']*1e3,2))"
done; done
