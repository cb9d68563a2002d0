#!/bin/bash
# gpurun_out/r04 (scratch, merged back from the GPU box) -> profiles/r04_* (tracked)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r04
for f in default driver_style_steps20 config2 config2_pinv config3_pinv config3b config3c config3l config4 config5; do
  [ -f $S/bench_$f.json ] && cp $S/bench_$f.json profiles/r04_bench_$f.json
done
for f in emulated_scaling_config4 emulated_scaling_config5; do [ -f $S/$f.json ] && cp $S/$f.json profiles/r04_$f.json; done
for f in strict_step interface_b accuracy_survey; do [ -f $S/$f.txt ] && cp $S/$f.txt profiles/r04_$f.txt; done
for f in $S/*_kernel_stats.csv $S/traffic_*.json $S/sq_counters_*.txt; do [ -f "$f" ] && cp "$f" profiles/r04_$(basename $f); done
ls profiles | grep r04_ | wc -l
