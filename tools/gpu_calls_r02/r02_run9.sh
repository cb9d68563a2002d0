#!/bin/bash
# round 2, GPU call: shader clock during the step kernel for both register caps (GRBM_GUI_ACTIVE = busy cycles of the
# dispatch; divided by its duration from the kernel trace = clock)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
for w in 2 4; do
RMP2_QUAD_MINW=$w rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/a -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > $O/bench_pmc_$w.json 2>/dev/null
echo "== RMP2_QUAD_MINW=$w"; python tools/pmc_sq.py $O/a; rm -rf $O/a
RMP2_QUAD_MINW=$w rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
find $O/k -name "*kernel_stats.csv" -exec head -2 {} \; | cut -c1-60,190-300; rm -rf $O/k
done
