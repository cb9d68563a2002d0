#!/bin/bash
# gpurun_out/r05/final (scratch, merged back from the GPU box by tools/refresh_profiles_r05{,_b}.sh) -> profiles/r05_* (tracked)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r05/final
for f in $S/bench_*.json $S/emulated_scaling_*.json $S/rehearsal_2ranks_*.json $S/fuzz_parity*.json $S/*_kernel_stats.csv $S/sq_counters_*.txt; do
  [ -f "$f" ] && cp "$f" profiles/r05_$(basename $f)
done
for f in accuracy_survey stamps dropin_latency; do [ -f $S/$f.txt ] && cp $S/$f.txt profiles/r05_$f.txt; done
for f in traffic_config2 traffic_config3 traffic_config3b executed_config3; do [ -f $S/$f.json ] && cp $S/$f.json profiles/$f.json; done
# interface B: the round's four measurement files in one
B=gpurun_out/r05
if [ -f $B/interface_b_stream.txt ]; then
  { echo "# interface B (config 3b: 256 explicit pairs per robot, 65 536 robots) -- round 5.  DESIGN.md section 8."
    echo "# (1) tools/gpu_calls_r05_*.sh: the two-phase streamed form (tools/experiments/r05_explicit_two_phase.patch) against the default"
    cat $B/interface_b_stream.txt
    echo; echo "# (2) its SQ counters against the default form's (rocprofv3 --pmc, own passes)"; cat $B/interface_b_stream_counters.txt
    echo; echo "# (3) the waves of a SIMD staggered by their slot (RMP2_STREAM_STAGGER, shader-clock units of 64)"; cat $B/interface_b_stagger.txt
    echo; echo "# (4) the single-loop streamed form in the tree (RMP2_EXPLICIT_STREAM=1) and its floors: pair arithmetic off / DMA off / both off"
    cat $B/interface_b_floor.txt
    echo; echo "# (5) the default two-wave form with the odd wave slot of every SIMD started late (first round only; RMP2_STREAM_STAGGER=n)"
    cat $B/interface_b_stagger_two_wave.txt; } | cut -c1-400 > profiles/r05_interface_b.txt
fi
ls profiles | grep r05_ | wc -l
