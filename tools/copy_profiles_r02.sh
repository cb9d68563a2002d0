#!/bin/bash
# copies what tools/refresh_profiles_r02.sh left under gpurun_out/r02 into the tracked profiles/ (names of profiles/README.md)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r02; O=profiles
KS=$(ls -t $S/kt3/*/*kernel_stats.csv | head -1)
cp $S/bench_default.json $O/r02_bench_default_config3.json
cp $S/bench_driver_style.json $O/r02_bench_driver_style_steps20.json
cp $S/bench_config2.json $O/r02_bench_config2.json
cp $S/bench_config2_64k.json $O/r02_bench_config2_64k.json
cp $S/bench_config3_4k.json $O/r02_bench_config3_4k.json
cp $S/bench_config4.json $O/r02_bench_config4_world1.json
cp $S/bench_config5.json $O/r02_bench_config5_world1.json
cp $S/bench_torchrun1.json $O/r02_bench_torchrun1.json
cp $KS $O/r02_config3_R65536_kernel_stats.csv
cp $S/traffic_config2.json $S/traffic_config3.json $O/
cp $S/ablation_R4096.txt $O/r02_ablation_R4096.txt
cp $S/ablation_R65536.txt $O/r02_ablation_R65536_after.txt
cp $S/sq_counters_quad_R65536_after.txt $O/r02_sq_counters_quad_R65536_after.txt
cp $S/sq_counters_hex_R4096.txt $O/r02_sq_counters_hex_R4096.txt
cp $S/stamps.txt $O/r02_stamps.txt
cp $S/rollout.txt $O/r02_rollout.txt
cp $S/exchange_timing.txt $O/r02_exchange_timing.txt
cp $S/pcie_inclusive.txt $O/r02_pcie_inclusive.txt
echo "copied; kernel stats from $KS"
