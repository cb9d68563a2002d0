#!/bin/bash
# round 3, GPU call 13: kernel timeline of the config-4 step, native and torch-driven exchange
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
for x in native torch; do
rocprofv3 --kernel-trace --output-format csv -d $O/kt_$x -- python3 bench.py --workload config4 --exchange $x --steps 200 --warmup 50 --no-cpu-baseline --no-secondary > /dev/null 2>&1
echo "== $x"; python tools/trace_timeline.py $O/kt_$x 600 14
done
rm -rf $O/kt_*/*/*.db
