#!/bin/bash
# round 5, first call: the GPU suite at the round's first milestone, the bench line on the reference's resolve, config 5 and the
# two-rank rehearsal (world-1 same-workload leg) with the new defaults
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite_a.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite_a.log; tail -8 $O/gpu_suite_a.log
timeout -k 10 600 python tools/accuracy_survey.py 2048 > $O/accuracy_survey_a.txt 2> $O/accuracy_survey_a.err || { tail -5 $O/accuracy_survey_a.err; }
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/bench_default_a.json 2> $O/bench_default_a.err || { tail -5 $O/bench_default_a.err; exit 1; }
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline --no-secondary > $O/bench_config5_a.json 2> $O/bench_config5_a.err || { tail -5 $O/bench_config5_a.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 2 --rehearse-one-gpu --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $O/rehearsal_config4_a.json 2> $O/rehearsal_config4_a.err || { tail -5 $O/rehearsal_config4_a.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_default_a", "bench_config5_a", "rehearsal_config4_a"):
    j = json.loads(open(f"gpurun_out/r05/{f}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f, round(j["ms_per_step"]*1e3, 2), "us", round(j["value"]/1e6, 1), "M/s", j["config"]["solve"], "frac", round(r["frac"], 3),
          "exec", r.get("executed_frac"), {k: (round(v["ms_per_step"]*1e3, 2) if isinstance(v, dict) and "ms_per_step" in v else None) for k, v in j.items() if k.startswith("solve_")},
          j.get("world1_same_workload_ms"), (j.get("world1_same_workload") or {}).get("error"))
PY
