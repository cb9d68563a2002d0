#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O
for lib in tools/diag/librmp2_sym0.so tools/diag/librmp2_sym1.so; do
RMP2_LIB=$PWD/$lib RMP2_KERNEL=quad timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or config5" 2>&1 | tail -1
for cfg in "65536 0" "65536 0" "49152 0" "262144 0" "65536 4" "16384 0"; do
set -- $cfg
RMP2_LIB=$PWD/$lib RMP2_QUAD_MINW=$2 timeout -k 10 120 python bench.py --robots $1 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$lib c3 $1 minw$2',round(j['ms_per_step']*1e3,2))"
done; done
