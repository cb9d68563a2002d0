#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
timeout -k 10 120 tools/diag/valu_rate > $O/valu_rate.txt; tail -8 $O/valu_rate.txt | cut -c1-100
rocprofv3 --pmc SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $O/a -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_sq.py $O/a; rm -rf $O/a
