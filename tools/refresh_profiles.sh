#!/bin/bash
# Regenerates gpurun_out/r01c/* (bench lines, rocprofv3 kernel stats, PMC traffic, ablations, stamps, rollout and
# launch-gap timings) in one go on a GPU box; the files judged are then copied into profiles/ (see profiles/README.md).
# Needs the diagnostic build next to the product library (built here, it travels with the snapshot):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared -Iinclude -DRMP2_STAMPS \
#         riemannian_motion_policies_amd/csrc/rmp2_hip.hip -o tools/diag/librmp2_stamps.so
# PMC passes are separate rocprofv3 runs (--pmc never combined with tracing), the program after "--" is python3 itself.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01c; mkdir -p $O
python bench.py > $O/bench_config2.json 2> $O/bench_config2.err
python bench.py --workload config3 > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --robots 65536 --no-cpu-baseline > $O/bench_config2_64k.json 2>/dev/null
python bench.py --workload config3 --robots 4096 --no-cpu-baseline > $O/bench_config3_4k.json 2>/dev/null
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/bench_config2_torchrun1.json 2>$O/torchrun.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --workload config3 --steps 100 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f2 -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w2 -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f3 -- python3 bench.py --workload config3 --steps 100 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w3 -- python3 bench.py --workload config3 --steps 100 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/f2 $O/w2 config2 4096 $O/traffic_config2.json
python tools/pmc_traffic.py $O/f3 $O/w3 config3 65536 $O/traffic_config3.json
for R in 4096 65536; do rocprofv3 --kernel-trace --output-format csv -d $O/abl_$R -- python3 tools/phase_timing.py $R > /dev/null 2>&1; python tools/trace_summary.py $O/abl_$R > $O/ablation_R$R.txt; done
RMP2_KERNEL=quad rocprofv3 --kernel-trace --output-format csv -d $O/ablq_4096 -- python3 tools/phase_timing.py 4096 > /dev/null 2>&1; python tools/trace_summary.py $O/ablq_4096 > $O/ablation_quad_R4096.txt
RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_stamps.so RMP2_KERNEL=hex python tools/stamps.py 4096 > $O/stamps_hex_R4096.txt 2>/dev/null
RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_stamps.so RMP2_KERNEL=quad python tools/stamps.py 4096 > $O/stamps_quad_R4096.txt 2>/dev/null
python tools/graph_gap.py config2 4096 200 > $O/graph_gap.txt 2>/dev/null
find $O -name "*kernel_stats.csv" | head; cat $O/bench_config2.json | cut -c1-300; cat $O/ablation_R4096.txt $O/stamps_hex_R4096.txt
python tools/rollout_timing.py config2 4096 50 > $O/rollout.txt 2>/dev/null
RMP2_KERNEL=quad python tools/rollout_timing.py config2 4096 50 >> $O/rollout.txt 2>/dev/null
python tools/rollout_timing.py config3 4096 50 >> $O/rollout.txt 2>/dev/null
RMP2_KERNEL=quad python tools/rollout_timing.py config3 4096 50 >> $O/rollout.txt 2>/dev/null
cat $O/rollout.txt
