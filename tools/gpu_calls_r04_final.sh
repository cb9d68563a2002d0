#!/bin/bash
# the whole GPU suite, then the round's evidence files (gpurun_out/r04/*: copied to profiles/ by tools/copy_profiles_r04.sh)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite.log; tail -6 $O/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
python tools/accuracy_survey.py 2048 > $O/accuracy_survey.txt 2>/dev/null || exit 1
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style_steps20.json 2> /dev/null || exit 1
for wl in config3b config3c config3l config2; do
  python bench.py --workload $wl --no-cpu-baseline --no-secondary > $O/bench_$wl.json 2> $O/bench_$wl.err || exit 1
done
python bench.py --workload config3 --solve pinv --no-cpu-baseline --no-secondary > $O/bench_config3_pinv.json 2> /dev/null || exit 1
python bench.py --workload config2 --solve pinv --no-cpu-baseline --no-secondary > $O/bench_config2_pinv.json 2> /dev/null || exit 1
python bench.py --workload config4 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2> /dev/null || exit 1
python bench.py --workload config5 --no-cpu-baseline --no-secondary > $O/bench_config5.json 2> /dev/null || exit 1
python bench.py --workload config4 --emulate-world 8 --steps 500 > $O/emulated_scaling_config4.json 2> /dev/null || exit 1
python bench.py --workload config5 --emulate-world 8 --steps 500 > $O/emulated_scaling_config5.json 2> /dev/null || exit 1
# strict step: certifying launch against the all-Jacobi two kernels
{
  echo "# solve = pinv (the reference's only resolve), config 3 / config 2: us per step, bench.py --solve pinv --steps 2000"
  for R in 65536 4096 64; do
    a=$(python bench.py --workload config3 --solve pinv --robots $R --no-cpu-baseline --no-secondary --steps 2000 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step']*1e3,2))")
    b=$(RMP2_STRICT_CERTIFY=0 python bench.py --workload config3 --solve pinv --robots $R --no-cpu-baseline --no-secondary --steps 1000 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step']*1e3,2))")
    c=$(python bench.py --workload config3 --robots $R --no-cpu-baseline --no-secondary --steps 2000 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step']*1e3,2))")
    echo "config3 R=$R: certifying one-launch step $a us | all-Jacobi two kernels (RMP2_STRICT_CERTIFY=0) $b us | solve=auto $c us"
  done
  python tools/diag_strict.py 65536 2>/dev/null | tail -6
} > $O/strict_step.txt
# interface B: register loads against the LDS-DMA stream, and what the memory system gives the access pattern
{
  echo "# interface B (config3b): us per step / fraction of 8 TB/s; g1 = RMP2_EXPLICIT_GLDS=1 (LDS-DMA stream + per-quad compaction)"
  for R in 32768 65536 131072; do for g in 0 1; do
    RMP2_EXPLICIT_GLDS=$g python bench.py --workload config3b --robots $R --no-cpu-baseline --no-secondary --steps 1000 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$R g$g:', round(j['ms_per_step']*1e3,2), 'us/step, hbm frac', round(j['roofline']['frac'],3))"
  done; done
  echo "# tools/stream_pairs.hip: the same access pattern without the control step (us per launch; 402.7 MB at R = 65536)"
  hipcc --offload-arch=gfx950 -O3 tools/stream_pairs.hip -o /tmp/stream_pairs 2>/dev/null && /tmp/stream_pairs 65536 | awk '{print $1, $2, $3, $4, $5, $6}'
} > $O/interface_b.txt
tail -3 $O/strict_step.txt; tail -12 $O/interface_b.txt
