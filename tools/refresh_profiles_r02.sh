#!/bin/bash
# Regenerates gpurun_out/r02/* in one go on a GPU box (bench lines, rocprofv3 kernel stats, PMC traffic, SQ counters,
# ablations, stamps, rollout / exchange timings); the files judged are then copied into profiles/ (profiles/README.md).
# Needs tools/diag/librmp2_stamps.so (-DRMP2_STAMPS build of the library) next to the product library.
# PMC passes are separate rocprofv3 runs (--pmc never combined with tracing); the program after "--" is python3 itself.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2>/dev/null
python bench.py --workload config2 --no-cpu-baseline > $O/bench_config2.json 2>/dev/null
python bench.py --workload config2 --robots 65536 --no-cpu-baseline > $O/bench_config2_64k.json 2>/dev/null
python bench.py --robots 4096 --no-cpu-baseline --no-secondary > $O/bench_config3_4k.json 2>/dev/null
python bench.py --workload config4 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2>$O/bench_config4.err
python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2>/dev/null
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun1.json 2>$O/torchrun.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --steps 200 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f2 -- python3 bench.py --workload config2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w2 -- python3 bench.py --workload config2 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3 $O/w3 config3 65536 $O/traffic_config3.json
python tools/pmc_traffic.py $O/f2 $O/w2 config2 4096 $O/traffic_config2.json
for R in 4096 65536; do
  rocprofv3 --kernel-trace --output-format csv -d $O/abl_$R -- python3 tools/phase_timing.py $R > /dev/null 2>&1; python tools/trace_summary.py $O/abl_$R > $O/ablation_R$R.txt
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 tools/phase_timing.py 65536 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/sq2 -- python3 tools/phase_timing.py 65536 > /dev/null 2>&1
(python tools/pmc_groups.py $O/sq1; python tools/pmc_groups.py $O/sq2) > $O/sq_counters_quad_R65536_after.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq3 -- python3 tools/phase_timing.py 4096 > /dev/null 2>&1
python tools/pmc_groups.py $O/sq3 > $O/sq_counters_hex_R4096.txt 2>&1
(RMP2_KERNEL=quad python tools/stamps.py 65536; RMP2_KERNEL=hex python tools/stamps.py 4096) > $O/stamps.txt 2>/dev/null
(python tools/rollout_timing.py config2 4096 50; RMP2_KERNEL=quad python tools/rollout_timing.py config2 4096 50; python tools/rollout_timing.py config3 4096 50; RMP2_KERNEL=quad python tools/rollout_timing.py config3 4096 50) > $O/rollout.txt 2>/dev/null
python tools/exchange_timing.py > $O/exchange_timing.txt 2>/dev/null
(python tools/pcie_inclusive.py config2 4096; python tools/pcie_inclusive.py config3 65536) > $O/pcie_inclusive.txt 2>/dev/null
rm -rf $O/kt3/*/*.db $O/f3 $O/w3 $O/f2 $O/w2 $O/abl_* $O/sq1 $O/sq2 $O/sq3
find $O -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-80,180-330
cut -c1-600 $O/bench_default.json; cat $O/traffic_config3.json $O/ablation_R65536.txt $O/sq_counters_quad_R65536_after.txt $O/stamps.txt $O/rollout.txt $O/exchange_timing.txt $O/pcie_inclusive.txt
du -sh $O
