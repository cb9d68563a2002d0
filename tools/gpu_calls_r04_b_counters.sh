#!/bin/bash
# interface B: what the SIMDs did -- SQ counters of the explicit-pair step (register loads / LDS-DMA stream with compaction) and of the
# shared-table step at the same two waves per SIMD
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
run() {  # name, env..., bench args
  local name=$1; shift
  env "$@" true
}
{
for cfg in "b_reg RMP2_EXPLICIT_GLDS=0 config3b" "b_dma RMP2_EXPLICIT_GLDS=1 config3b" "shared_2waves RMP2_QUAD_MINW=2 config3"; do
  set -- $cfg
  echo "== $1 ($2, --workload $3)"
  export $2
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/c1_$1 -- python3 bench.py --workload $3 --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/c2_$1 -- python3 bench.py --workload $3 --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python tools/pmc_sq.py $O/c1_$1; python tools/pmc_sq.py $O/c2_$1
  unset ${2%%=*}
  rm -rf $O/c1_$1 $O/c2_$1
done
} > $O/interface_b_counters.txt 2>&1
cat $O/interface_b_counters.txt
