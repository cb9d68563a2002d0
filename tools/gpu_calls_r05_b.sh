#!/bin/bash
# round 5, second call: the suite with the refined norm and the envelope gate, then the survey
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/gpu_suite_b.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/gpu_suite_b.log; tail -12 $O/gpu_suite_b.log
timeout -k 10 600 python tools/accuracy_survey.py 2048 > $O/accuracy_survey_b.txt 2> $O/accuracy_survey_b.err || { tail -5 $O/accuracy_survey_b.err; }
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_default_b.json 2> $O/bench_default_b.err || { tail -5 $O/bench_default_b.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05/bench_default_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(round(j["ms_per_step"]*1e3, 2), "us", round(j["value"]/1e6, 1), "M/s", j["config"]["solve"], "frac", round(r["frac"], 3), "exec", r.get("executed_frac"), j["result_check"]["admitted_by"], j["result_check"].get("passing_each_clause_on_its_own"))
print({k: round(v["ms_per_step"]*1e3, 2) for k, v in j.items() if k.startswith("solve_")}, j["secondary"]["ms_per_step"]*1e3)
PY
