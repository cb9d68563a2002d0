#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -E "^\s*(Name|Counter).*(SQ_|SQC_)" $O/counters.txt | head -300 > $O/sq_list.txt; wc -l $O/sq_list.txt
for V in 4 2; do
export RMP2_QUAD_MINW=$V
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/a$V -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/b$V -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/c$V -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
for d in a b c; do echo "== minw $V pass $d"; python tools/pmc_sq.py $O/$d$V 2>&1 | tail -12; done
done > $O/sq_ab.txt 2>&1
cat $O/sq_ab.txt
rm -rf $O/a? $O/b? $O/c?
