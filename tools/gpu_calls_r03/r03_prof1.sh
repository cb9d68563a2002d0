#!/bin/bash
# round 3, profile pass 1 (mid-round): kernel trace, HBM traffic counters, SQ counters, stamps of config 3 at 65 536 robots
# with the per-mode plain builds; interface B traffic
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p1; mkdir -p $O
C=${COMMIT:-unknown}
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --steps 200 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3 $O/w3 config3 65536 $O/traffic_config3.json $C
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3b $O/w3b config3b 65536 $O/traffic_config3b.json $C
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/sq2 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
(python tools/pmc_sq.py $O/sq1; python tools/pmc_sq.py $O/sq2) > $O/sq_counters_config3_R65536.txt 2>&1
RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_stamps.so RMP2_KERNEL=quad python tools/stamps.py 65536 > $O/stamps.txt 2> $O/stamps.err
python tools/executed.py $O/sq1 $O/sq2 $O/stamps.txt config3 65536 $O/executed_config3.json $C
rm -rf $O/kt3/*/*.db $O/f3 $O/w3 $O/f3b $O/w3b $O/sq1 $O/sq2
find $O -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-80,180-330
cat $O/traffic_config3.json $O/traffic_config3b.json $O/sq_counters_config3_R65536.txt $O/stamps.txt
du -sh $O
