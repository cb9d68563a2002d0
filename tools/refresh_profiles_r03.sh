#!/bin/bash
# Regenerates gpurun_out/r03/* in one go on a GPU box (bench lines, rocprofv3 kernel stats, PMC traffic, SQ counters, stamps,
# sweeps, emulated scaling, exchange / rollout timings); tools/copy_profiles_r03.sh then copies the files judged into profiles/.
# Needs tools/diag/librmp2_stamps.so (python -c "import __graft_entry__ as g; g.build_hip(variant='stamps',
# defines=['-DRMP2_STAMPS', '-DRMP2_TUNING'])").  PMC passes are separate rocprofv3 runs (--pmc never combined with
# tracing); the program after "--" is python3 itself.  COMMIT = the commit of the snapshot (the GPU box has no .git).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
C=${COMMIT:-unknown}
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2>/dev/null
python bench.py --solve pinv --no-cpu-baseline --no-secondary > $O/bench_config3_pinv.json 2>/dev/null
python bench.py --workload config2 --no-cpu-baseline > $O/bench_config2.json 2>/dev/null
python bench.py --workload config3b --no-cpu-baseline > $O/bench_config3b.json 2>/dev/null
python bench.py --workload config3c --no-cpu-baseline --no-secondary > $O/bench_config3c.json 2>/dev/null
python bench.py --workload config3l --no-cpu-baseline --no-secondary > $O/bench_config3l.json 2>$O/bench_config3l.err
python bench.py --workload config4 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2>$O/bench_config4.err
python bench.py --workload config4 --exchange torch --no-cpu-baseline --no-secondary > $O/bench_config4_torch.json 2>/dev/null
python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2>/dev/null
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun1.json 2>$O/torchrun.err
python bench.py --workload config5 --emulate-world 8 --compare-flop-model > $O/emulated_scaling_config5.json 2>/dev/null
python bench.py --workload config4 --emulate-world 8 > $O/emulated_scaling_config4.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --steps 200 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -- python3 bench.py --workload config5 --steps 200 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3 $O/w3 config3 65536 $O/traffic_config3.json $C
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3b $O/w3b config3b 65536 $O/traffic_config3b.json $C
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f2 -- python3 bench.py --workload config2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w2 -- python3 bench.py --workload config2 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/f2 $O/w2 config2 4096 $O/traffic_config2.json $C
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq2 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
(python tools/pmc_sq.py $O/sq1; python tools/pmc_sq.py $O/sq2) > $O/sq_counters_config3_R65536.txt 2>&1
(RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_stamps.so RMP2_KERNEL=quad python tools/stamps.py 65536; RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_stamps.so RMP2_KERNEL=hex python tools/stamps.py 4096) > $O/stamps.txt 2> $O/stamps.err
python tools/executed.py $O/sq1 $O/sq2 $O/stamps.txt config3 65536 $O/executed_config3.json $C
python tools/minw_sweep.py config3 > $O/quad_minw_ab.txt 2>/dev/null
(python tools/dispatch_sweep.py tj5; python tools/dispatch_sweep.py config3r 4096 8192 12288 16384 20480 32768; python tools/dispatch_sweep.py config3 4096 8192 12288 16384 32768; python tools/dispatch_sweep.py config2 4096 8192 16384 32768 65536) > $O/dispatch_sweep.txt 2>/dev/null
python tools/calibrate_costs.py > $O/cost_calibration.json 2>/dev/null
(python tools/rollout_timing.py config2 4096 50; python tools/rollout_timing.py config3 4096 50; RMP2_KERNEL=quad python tools/rollout_timing.py config3 65536 20) > $O/rollout.txt 2>/dev/null
python tools/fence_cost.py > $O/exchange_timing.txt 2>/dev/null
(python tools/pcie_inclusive.py config2 4096; python tools/pcie_inclusive.py config3 65536) > $O/pcie_inclusive.txt 2>/dev/null
python tools/flag_tail.py > $O/flag_tail.txt 2>/dev/null
python tools/closest_stage_timing.py 65536 50 > $O/closest_stage.txt 2>/dev/null
python tools/rollout_diag.py 65536 > $O/rollout_diag.txt 2>/dev/null
(python tools/dropin_latency.py 7 300 65536; python tools/dropin_latency.py 32 300 65536) > $O/dropin_latency.txt 2>/dev/null
rm -rf $O/kt*/*/*.db $O/f3 $O/w3 $O/f3b $O/w3b $O/f2 $O/w2 $O/sq1 $O/sq2
# the contract line once more, now that the counter files of THESE kernels exist (the line then carries traffic + executed)
cp $O/traffic_config2.json $O/traffic_config3.json $O/traffic_config3b.json $O/executed_config3.json profiles/
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --workload config3b --no-cpu-baseline > $O/bench_config3b.json 2>/dev/null
(python tools/dispatch_sweep.py exp05 4096 16384 20480 32768 65536; python tools/dispatch_sweep.py exp05tj 4096 16384 20480 32768 65536; python tools/minw_sweep.py config3j 32768 49152 57344 65536 98304 131072 262144) >> $O/dispatch_sweep.txt 2>/dev/null
find $O -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-100,180-330
cut -c1-400 $O/bench_default.json; cat $O/traffic_config3.json $O/executed_config3.json $O/stamps.txt $O/rollout.txt $O/exchange_timing.txt $O/pcie_inclusive.txt
du -sh $O
