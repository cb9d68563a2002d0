#!/bin/bash
# Round-5 evidence, part 1 (one GPU call): the whole GPU suite, the accuracy survey, the bench lines, the two-rank rehearsal.
# Part 2: tools/refresh_profiles_r05_b.sh (rocprofv3 kernel stats, PMC traffic, SQ counters + stamps, the fuzz campaign).
# Both write under gpurun_out/r05/final/; tools/copy_profiles_r05.sh copies the judged files into profiles/.
O=gpurun_out/r05/final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite.log; tail -6 $O/gpu_suite.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/accuracy_survey.py 2048 $O/accuracy_survey > $O/accuracy_survey.txt 2> $O/accuracy_survey.err || { tail -5 $O/accuracy_survey.err; exit 1; }
tail -16 $O/accuracy_survey.txt | cut -c1-330
(python tools/dropin_latency.py 7 300; python tools/dropin_latency.py 32 300) > $O/dropin_latency.txt 2> /dev/null || true
python __graft_entry__.py smoke > $O/smoke.txt 2>&1 || { tail -3 $O/smoke.txt; exit 1; }
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python bench.py --steps 20 --warmup 5 > $O/bench_driver_style_steps20.json 2> /dev/null || exit 1
for wl in config3b config3c config3l config2 config4 config5; do
  python bench.py --workload $wl --no-cpu-baseline --no-secondary > $O/bench_$wl.json 2> $O/bench_$wl.err || { tail -5 $O/bench_$wl.err; exit 1; }
done
python bench.py --workload config3 --solve auto --no-cpu-baseline --no-secondary > $O/bench_config3_auto.json 2> /dev/null || exit 1
python bench.py --workload config5 --link-geometry --no-cpu-baseline > $O/bench_config5_link_geometry.json 2> /dev/null || exit 1
python bench.py --workload config4 --emulate-world 8 --no-cpu-baseline > $O/emulated_scaling_config4.json 2> /dev/null || exit 1
python bench.py --workload config5 --emulate-world 8 --no-cpu-baseline > $O/emulated_scaling_config5.json 2> /dev/null || exit 1
timeout -k 10 400 python bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --rank-timeout 300 --no-cpu-baseline --no-secondary > $O/rehearsal_2ranks_config4.json 2> $O/rehearsal_2ranks_config4.err || { tail -5 $O/rehearsal_2ranks_config4.err; exit 1; }
timeout -k 10 400 python bench.py --gpus 2 --rehearse-one-gpu --workload config5 --steps 200 --warmup 20 --rank-timeout 300 --no-cpu-baseline > $O/rehearsal_2ranks_config5.json 2> $O/rehearsal_2ranks_config5.err || { tail -5 $O/rehearsal_2ranks_config5.err; exit 1; }
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $O/rehearsal_2ranks_config4_torchrun.json 2> $O/rehearsal_2ranks_config4_torchrun.err || { tail -5 $O/rehearsal_2ranks_config4_torchrun.err; exit 1; }
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/final/bench_*.json")) + sorted(glob.glob("gpurun_out/r05/final/rehearsal_2ranks_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
    print(f.split("/")[-1][:-5].ljust(34), f"{j['ms_per_step']*1e3:8.2f} us  {j['value']/1e6:8.1f} M/s  {str(j['config'].get('solve')):5s} {r['bound']:4s} frac {r['frac']:.3f}",
          "exec", None if r.get("executed_frac") is None else round(r["executed_frac"], 3),
          {k: round(v["ms_per_step"]*1e3, 2) for k, v in j.items() if k.startswith("solve_")}, j.get("world1_same_workload_ms"))
PY
