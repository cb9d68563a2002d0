#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/strict2_tests.log 2>&1 || { tail -50 gpurun_out/strict2_tests.log; exit 1; }
tail -2 gpurun_out/strict2_tests.log
python bench.py --solve pinv --no-cpu-baseline --no-secondary > gpurun_out/bench_config3_pinv.json 2> gpurun_out/bench_config3_pinv.err || { tail -20 gpurun_out/bench_config3_pinv.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_config3_pinv.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d.get("kernel"), d["result_check"])
PY
