#!/bin/bash
# whole GPU suite + smoke + default bench line
set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/full_suite.log 2>&1 || { tail -40 gpurun_out/full_suite.log; exit 1; }
tail -3 gpurun_out/full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"].get("traffic"))
PY
