#!/bin/bash
# copies what tools/refresh_profiles_r03.sh left under gpurun_out/r03 into the tracked profiles/ (names of profiles/README.md)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r03; O=profiles
cp $S/bench_default.json $O/r03_bench_default_config3.json
cp $S/bench_driver_style.json $O/r03_bench_driver_style_steps20.json
cp $S/bench_config3_pinv.json $O/r03_bench_config3_pinv.json
cp $S/bench_config2.json $O/r03_bench_config2.json
cp $S/bench_config3b.json $O/r03_bench_config3b_explicit_pairs.json
cp $S/bench_config3c.json $O/r03_bench_config3c_capsules.json
cp $S/bench_config3l.json $O/r03_bench_config3l_link_geometry.json
cp $S/bench_config4.json $O/r03_bench_config4_world1.json
cp $S/bench_config4_torch.json $O/r03_bench_config4_world1_torch_exchange.json
cp $S/bench_config5.json $O/r03_bench_config5_world1.json
cp $S/bench_torchrun1.json $O/r03_bench_torchrun1.json
cp $S/emulated_scaling_config5.json $O/r03_emulated_scaling.json
cp $S/emulated_scaling_config4.json $O/r03_emulated_scaling_config4.json
cp $(ls -t $S/kt3/*/*kernel_stats.csv | head -1) $O/r03_config3_R65536_kernel_stats.csv
cp $(ls -t $S/kt3b/*/*kernel_stats.csv | head -1) $O/r03_config3b_R65536_kernel_stats.csv
cp $(ls -t $S/kt5/*/*kernel_stats.csv | head -1) $O/r03_config5_kernel_stats.csv
cp $S/traffic_config2.json $S/traffic_config3.json $S/traffic_config3b.json $S/executed_config3.json $O/
cp $S/sq_counters_config3_R65536.txt $O/r03_sq_counters_config3_R65536.txt
cp $S/stamps.txt $O/r03_stamps.txt
cp $S/quad_minw_ab.txt $O/r03_quad_minw_ab.txt
cp $S/dispatch_sweep.txt $O/r03_dispatch_sweep.txt
cp $S/cost_calibration.json $O/r03_cost_calibration.json
cp $S/rollout.txt $O/r03_rollout.txt
cp $S/exchange_timing.txt $O/r03_exchange_timing.txt
cp $S/pcie_inclusive.txt $O/r03_pcie_inclusive.txt
cp $S/flag_tail.txt $O/r03_flag_tail.txt
# (these three keep the hand-written header lines of the tracked file: "# ..." lines at its top)
for f in closest_stage rollout_diag dropin_latency; do
  { grep '^#' $O/r03_$f.txt || true; grep -v 'amdgpu.ids' $S/$f.txt; } > $O/r03_$f.txt.new && mv $O/r03_$f.txt.new $O/r03_$f.txt
done
echo copied
