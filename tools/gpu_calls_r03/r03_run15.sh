#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt_native -- python3 bench.py --workload config4 --exchange native --steps 300 --warmup 50 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/trace_gaps.py $O/kt_native
rm -rf $O/kt_*/*/*.db
