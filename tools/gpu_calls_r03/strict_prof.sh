#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/strict_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/strict_prof -- python3 bench.py --solve pinv --steps 200 --no-cpu-baseline --no-secondary > /dev/null 2>&1
f=$(ls -t gpurun_out/strict_prof/*/*kernel_stats.csv | head -1)
head -6 $f | cut -c1-60,150-330
rm -f gpurun_out/strict_prof/*/*.db
