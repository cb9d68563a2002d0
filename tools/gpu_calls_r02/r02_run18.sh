#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_random_robots.py tests/test_gpu_dropin.py -m gpu -x -q > $O/pytest_gpu_quad.txt 2>&1; tail -3 $O/pytest_gpu_quad.txt
for lib in tools/diag/librmp2_nopk.so riemannian_motion_policies_amd/librmp2_hip.so; do
for cfg in "65536 2" "65536 2" "49152 3" "262144 3" "32768 2"; do
set -- $cfg
RMP2_LIB=$PWD/$lib RMP2_QUAD_MINW=$2 timeout -k 10 120 python bench.py --robots $1 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$lib c3 $1 minw$2',j['ms_per_step'])"
done; done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/a -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_sq.py $O/a; rm -rf $O/a
