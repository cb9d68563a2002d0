#!/bin/bash
# round 2, GPU call 1: instruction-rate microbenchmark, parity suite, the new bench lines, "before" profiles
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout -k 10 120 tools/diag/valu_rate > $O/valu_rate.txt 2>&1; tail -40 $O/valu_rate.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-1500 $O/bench_default.json; tail -3 $O/bench_default.err
timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_style.json 2>/dev/null; cut -c1-400 $O/bench_driver_style.json
timeout -k 10 60 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?"; tail -2 $O/bench_gpus2.err
timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-600 $O/bench_config4.json; tail -3 $O/bench_config4.err
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-900 $O/bench_config5.json; tail -3 $O/bench_config5.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --no-secondary > $O/bench_torchrun1.json 2>$O/torchrun.err; cut -c1-300 $O/bench_torchrun1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --steps 200 --no-cpu-baseline > /dev/null 2>&1
find $O/kt3 -name "*kernel_stats.csv" -exec head -5 {} \;
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 tools/phase_timing.py 65536 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/sq2 -- python3 tools/phase_timing.py 65536 > /dev/null 2>&1
(python tools/pmc_groups.py $O/sq1; python tools/pmc_groups.py $O/sq2) > $O/sq_counters_R65536_before.txt 2>&1; cat $O/sq_counters_R65536_before.txt
rocprofv3 --kernel-trace --output-format csv -d $O/abl -- python3 tools/phase_timing.py 65536 > /dev/null 2>&1; python tools/trace_summary.py $O/abl > $O/ablation_R65536_before.txt; cat $O/ablation_R65536_before.txt
rm -rf $O/kt3/*/*.db $O/sq1 $O/sq2 $O/abl 2>/dev/null; du -sh $O
