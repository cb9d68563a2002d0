#!/bin/bash
# Round-5 evidence, part 2: rocprofv3 kernel-trace stats of the timed workloads, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes;
# gfx950 correction in tools/pmc_traffic.py), SQ counters + stamps of the headline kernel (executed_config3.json: everything in it measured
# at the kernel hash it carries), the fuzz campaign.  The program after "--" is python3 itself; --pmc is never combined with tracing.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/final; mkdir -p $O
C=${COMMIT:-unknown}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --steps 200 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
cp $(ls -t $O/kt3/*/*kernel_stats.csv | head -1) $O/config3_R65536_kernel_stats.csv
cp $(ls -t $O/kt3b/*/*kernel_stats.csv | head -1) $O/config3b_R65536_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3 $O/w3 config3 65536 $O/traffic_config3.json $C
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3b -- python3 bench.py --workload config3b --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/pmc_traffic.py $O/f3b $O/w3b config3b 65536 $O/traffic_config3b.json $C
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f2 -- python3 bench.py --workload config2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w2 -- python3 bench.py --workload config2 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/f2 $O/w2 config2 4096 $O/traffic_config2.json $C
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq2 -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
(python tools/pmc_sq.py $O/sq1; python tools/pmc_sq.py $O/sq2) > $O/sq_counters_config3_R65536.txt 2>&1
# in-kernel stamps of THIS round's kernels (tools/diag/librmp2_stamps.so: -DRMP2_STAMPS build of the same sources)
(RMP2_KERNEL=quad python tools/stamps.py 65536; RMP2_KERNEL=hex python tools/stamps.py 4096) > $O/stamps.txt 2>/dev/null
python tools/executed.py $O/sq1 $O/sq2 $O/stamps.txt config3 65536 $O/executed_config3.json $C
rm -rf $O/kt*/*/*.db $O/f3 $O/w3 $O/f3b $O/w3b $O/f2 $O/w2 $O/sq1 $O/sq2
cp $O/traffic_config2.json $O/traffic_config3.json $O/traffic_config3b.json $O/executed_config3.json profiles/
# the contract line once more, now that the counter files of THESE kernels exist (the line then carries traffic + executed)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --workload config3b --no-cpu-baseline --no-secondary > $O/bench_config3b.json 2>/dev/null
for f in $O/*kernel_stats.csv; do echo $f; head -3 $f | cut -c1-100,180-330; done
cat $O/traffic_config3.json $O/traffic_config3b.json $O/executed_config3.json | cut -c1-200; cat $O/stamps.txt | cut -c1-250
timeout -k 10 560 python tools/fuzz_parity.py --seeds 2000000 2060000 --minutes 8 --log $O/fuzz_parity.log > $O/fuzz_parity.json 2>&1; tail -42 $O/fuzz_parity.json | cut -c1-200
timeout -k 10 100 python tools/fuzz_parity.py --pairs --seeds 2000 2400 --minutes 1.2 > $O/fuzz_parity_pair_grid.json 2>&1; tail -12 $O/fuzz_parity_pair_grid.json | cut -c1-200
