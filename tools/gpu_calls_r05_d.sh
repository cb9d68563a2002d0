#!/bin/bash
# interface B, streamed form: SQ counters next to the single-loop form's and the shared-table four-wave step's
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
{
for cfg in "b_single RMP2_EXPLICIT_STREAM=0 config3b" "b_stream RMP2_EXPLICIT_STREAM=1 config3b" "shared_4waves RMP2_QUAD_MINW=4 config3"; do
  set -- $cfg
  echo "== $1 ($2, --workload $3)"
  export $2
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/c1_$1 -- python3 bench.py --workload $3 --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/c2_$1 -- python3 bench.py --workload $3 --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/c3_$1 -- python3 bench.py --workload $3 --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python tools/pmc_sq.py $O/c1_$1; python tools/pmc_sq.py $O/c2_$1; python tools/pmc_sq.py $O/c3_$1
  unset ${2%%=*}
  rm -rf $O/c1_$1 $O/c2_$1 $O/c3_$1
done
} > $O/interface_b_stream_counters.txt 2>&1
cat $O/interface_b_stream_counters.txt | cut -c1-200
