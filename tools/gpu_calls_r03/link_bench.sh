#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/dropin_tests.log 2>&1 || { tail -40 gpurun_out/dropin_tests.log; exit 1; }
tail -2 gpurun_out/dropin_tests.log
python bench.py --workload config3l --no-cpu-baseline --no-secondary > gpurun_out/bench_config3l.json 2> gpurun_out/bench_config3l.err || { tail -20 gpurun_out/bench_config3l.err; exit 1; }
cut -c1-600 gpurun_out/bench_config3l.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_config3l.json").read().strip().splitlines()[-1])
print(d["result_check"], d["roofline"]["frac"], d.get("kernel"))
PY
