#!/bin/bash
# 2-dof robots, solve = pinv without a certificate, small fleets: quad (closed-form 2 x 2 pseudo-inverse) instead of hex (careful path)
O=gpurun_out/r05/final; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q > gpurun_out/r05/gpu_suite_cal2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/gpu_suite_cal2.log; tail -3 gpurun_out/r05/gpu_suite_cal2.log | cut -c1-200
python tools/calibrate_costs.py pinv > $O/cost_calibration.json 2> $O/cost_calibration.err || { tail -3 $O/cost_calibration.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05/final/cost_calibration.json"))
for k, v in j["curves"].items(): print(k, v["robots"][:8], v["us"])
PY
python bench.py --workload config5 --emulate-world 8 --no-cpu-baseline > $O/emulated_scaling_config5.json 2> $O/emulated_scaling_config5.err
python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2>/dev/null
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05/final/emulated_scaling_config5.json").read().strip().splitlines()[-1])["emulated_scaling"]
print({k: j[k] for k in ("max_us", "mean_us", "imbalance_max_over_mean", "predicted_8gpu_steps_per_s")})
j = json.loads(open("gpurun_out/r05/final/bench_config5.json").read().strip().splitlines()[-1]); print("config5", j["ms_per_step"] * 1e3)
PY
